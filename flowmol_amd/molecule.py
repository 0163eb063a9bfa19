"""``SampledMolecule``: the result object of ``FlowMol.sample*`` (drop-in for the fields of reference
flowmol/analysis/molecule_builder.py:17-84,217-265 that do not need RDKit).

The reference keeps float one-hots on a per-molecule DGL graph; here a molecule is a few index
tensors, so trajectories cost bytes instead of megabytes.  ``rdkit_mol`` is
built lazily and is ``None`` when RDKit is not installed (it is not, in this image).
"""
from __future__ import annotations

import functools
from typing import Dict, List, Optional

import numpy as np
import torch


@functools.lru_cache(maxsize=256)
def pair_indices(n: int):
    """(src, dst) of the unordered pairs in the reference's upper-edge order
    (torch.triu_indices(n, n, 1), flowmol/data_processing/utils.py:4-17).  Cached per size: packaging a batch
    calls this once per molecule."""
    up = torch.triu_indices(n, n, offset=1)
    return up[0], up[1]


def extract_moldata(x, a_idx, c_idx, e_idx, atom_type_map: List[str], fake_atoms: bool, n_bond_types: int = 4,
                    show_fake_atoms: bool = False):
    """Tokens -> (positions, symbols, charges, bond_types, bond_src, bond_dst).

    Semantics of reference extract_moldata_from_graph (molecule_builder.py:217-265): atoms whose type is
    the fake-atom index (= len(atom_type_map), before the mask token) are removed and bonds re-indexed
    (DGL remove_nodes); charge = index - 2; bond order 0 and the mask index mean "no bond"; only
    upper-triangle bonds are kept."""
    n = int(a_idx.shape[0])
    amap = list(atom_type_map) + (['Sn'] if fake_atoms else []) + ['Se']      # molecule_builder.py:40-44
    keep = torch.ones(n, dtype=torch.bool)
    if fake_atoms and not show_fake_atoms:
        keep = a_idx != len(atom_type_map)
    new_id = torch.cumsum(keep.long(), 0) - 1
    src, dst = pair_indices(n)
    bt = e_idx.clone().long()
    bt[bt == n_bond_types] = 0
    ok = keep[src] & keep[dst] & (bt != 0)
    symbols = [amap[int(i)] for i in a_idx[keep]]
    return x[keep], symbols, c_idx[keep].long() - 2, bt[ok], new_id[src[ok]], new_id[dst[ok]]


class SampledMolecule:
    """One sampled molecule.  Fields follow the reference's SampledMolecule."""

    def __init__(self, x_1: torch.Tensor, a_1: torch.Tensor, c_1: torch.Tensor, e_1: torch.Tensor,
                 atom_type_map: List[str], fake_atoms: bool = False, ctmc_mol: bool = True,
                 explicit_aromaticity: bool = False, traj_frames: Optional[Dict[str, torch.Tensor]] = None,
                 build_xt_traj: bool = True, build_ep_traj: bool = True, align_traj: bool = True, n_charges: int = 6):
        self.atom_type_map_in = list(atom_type_map)
        self.fake_atoms = fake_atoms
        self.ctmc_mol = ctmc_mol
        self.explicit_aromaticity = explicit_aromaticity
        self.n_bond_types = 5 if explicit_aromaticity else 4
        self.n_charges = n_charges
        # raw final state (tokens), kept like the reference keeps ``self.g``
        self.x_1, self.a_1, self.c_1, self.e_1 = x_1, a_1.long(), c_1.long(), e_1.long()
        (self.positions, self.atom_types, self.atom_charges, self.bond_types, self.bond_src_idxs,
         self.bond_dst_idxs) = extract_moldata(x_1, self.a_1, self.c_1, self.e_1, atom_type_map, fake_atoms, self.n_bond_types)
        self.atom_type_map = list(atom_type_map) + (['Sn'] if fake_atoms else []) + (['Se'] if ctmc_mol else [])
        self.num_atom_types = len(self.atom_type_map)
        self.num_atoms = int(x_1.shape[0])
        if fake_atoms:
            self.num_atoms -= int((self.a_1 == len(atom_type_map)).sum())
        self.valencies = self.compute_valencies(arom_dependent=explicit_aromaticity)
        self.traj_frames = traj_frames
        self._rdkit_mol = False      # lazily built
        self.build_xt_traj, self.build_ep_traj, self.align_traj = build_xt_traj, build_ep_traj, align_traj
        self._traj_mols: Dict[bool, list] = {}

    def compute_valencies(self, arom_dependent: bool = False):
        """molecule_builder.py:138-157."""
        adj = torch.zeros((self.num_atoms, self.num_atoms)).float()
        bt = self.bond_types.clone().float()
        bt[bt == 4] = 1.5
        adj[self.bond_src_idxs, self.bond_dst_idxs] = bt
        adj[self.bond_dst_idxs, self.bond_src_idxs] = bt
        val = torch.sum(adj, dim=-1)
        if arom_dependent:
            n_arom = (adj == 1.5).sum(dim=-1)
            val = torch.stack([n_arom, (val - n_arom * 1.5).long()], dim=1)
        return val

    # ---------------------------------------------------------------- optional RDKit
    @property
    def rdkit_mol(self):
        if self._rdkit_mol is False:
            self._rdkit_mol = build_rdkit_mol(self.positions, self.atom_types, self.atom_charges, self.bond_src_idxs,
                                              self.bond_dst_idxs, self.bond_types)
        return self._rdkit_mol

    # ---------------------------------------------------------------- trajectories as RDKit molecules (molecule_builder.py:76-84,156-214)
    def _traj_mols_of(self, ep_traj: bool):
        """The reference builds these lists eagerly in __init__ (`self.traj_mols = self.process_traj_frames(traj_frames)` when
        build_xt_traj, `self.ep_traj_mols = ...` when build_ep_traj and the frames hold 'x_1_pred'); its caller iterates them into an
        RDKit SDWriter (test.py:235,251).  Here they are built on first access -- a 500-step trajectory is 501 RDKit molecules per
        sampled molecule -- with the same switches: the attribute does not exist (AttributeError) when the molecule carries no frames
        or the matching build switch is off.  They ARE RDKit molecules, so RDKit must be importable; without it the RDKit-free
        equivalent is ``traj_mol_blocks(ep_traj)`` (the same frames, aligned the same way, as V2000 mol blocks)."""
        name = 'ep_traj_mols' if ep_traj else 'traj_mols'
        built = self.build_ep_traj if ep_traj else self.build_xt_traj
        if self.traj_frames is None or not built or (ep_traj and 'x_1_pred' not in self.traj_frames):
            raise AttributeError(f"'SampledMolecule' object has no attribute {name!r} (sample with {'ep_traj' if ep_traj else 'xt_traj'}=True)")
        if ep_traj not in self._traj_mols:
            try:
                import rdkit       # noqa: F401
            except Exception as e:
                raise ImportError(f'SampledMolecule.{name} is a list of RDKit molecules and RDKit is not installed; '
                                  f'use SampledMolecule.traj_mol_blocks(ep_traj={ep_traj}) for the same frames as V2000 mol blocks') from e
            tf = self.traj_frames
            sfx = '_1_pred' if ep_traj else ''
            n_frames = int(tf['x' + sfx].shape[0])
            x_final = tf['x' + sfx][-1]
            mols = []
            for f in range(n_frames):
                pos, sym, chg, bt, bs, bd = self.frame_moldata(f, ep_traj)
                if self.align_traj:
                    pos = rigid_alignment(pos, x_final)
                mols.append(build_rdkit_mol(pos, sym, chg, bs, bd, bt))
            kept = [m for m in mols if m is not None]        # molecule_builder.py:208
            if len(kept) < n_frames:
                print(f'WARNING: {n_frames - len(kept)} frames were not converted to rdkit molecules')
            self._traj_mols[ep_traj] = kept
        return self._traj_mols[ep_traj]

    @property
    def traj_mols(self):
        return self._traj_mols_of(False)

    @property
    def ep_traj_mols(self):
        return self._traj_mols_of(True)

    def traj_frames_reference(self) -> Dict[str, torch.Tensor]:
        """The molecule's trajectory frames in the REFERENCE's format (ctmc_vector_field.py:188-202,235-255,267-283), rebuilt on demand from the
        compact frames this build keeps (``traj_frames``: int tokens per atom / per unordered pair): keys 'x', 'a', 'c', 'e' with T frames
        (frame 0 = the prior) and 'x_1_pred', 'a_1_pred', 'c_1_pred', 'e_1_pred' with T - 1; categorical frames are float one-hots
        INCLUDING the mask column -- (frames, n, n_atom_types + 1), (frames, n, n_charges + 1) -- and edge frames cover ALL n (n - 1) directed
        edges in the reference's order, the upper-triangle pairs followed by the same pairs swapped (data_processing/utils.py:4-17; both
        directions carry the same token, ctmc_vector_field.py:397-409): (frames, n (n - 1), n_bond_types + 1).  A caller written against the
        reference's ``SampledMolecule.traj_frames`` reads this instead; nothing is materialised until it is called (the compact form of a
        60-atom, 500-step trajectory is 2.4 MB, the reference's 35 MB).  CTMC models with the default `campbell` integrator: a `gat` run's
        reference '*_1_pred' frames hold tempered probabilities (ctmc_vector_field.py:373-375), of which the compact frames keep the argmax
        only, and an endpoint-parameterised model's frames are continuous vectors -- neither can be rebuilt."""
        if self.traj_frames is None:
            raise AttributeError("'SampledMolecule' object has no trajectory frames (sample with xt_traj=True or ep_traj=True)")
        if not self.ctmc_mol:
            raise NotImplementedError("endpoint-parameterised models keep the argmax of their continuous frames only: the reference's float frames cannot be rebuilt")
        one_hot = torch.nn.functional.one_hot
        tf = self.traj_frames
        widths = {'a': len(self.atom_type_map_in) + (1 if self.fake_atoms else 0) + 1, 'c': self.n_charges + 1, 'e': self.n_bond_types + 1}
        out = {}
        for sfx in ('', '_1_pred'):
            out['x' + sfx] = tf['x' + sfx].detach().to(torch.float32).cpu()
            for k in 'ac':
                out[k + sfx] = one_hot(tf[k + sfx].detach().cpu().long(), widths[k]).float()
            e = tf['e' + sfx].detach().cpu().long()
            out['e' + sfx] = one_hot(torch.cat([e, e], dim=1), widths['e']).float()
        return out

    def frame_moldata(self, frame_idx: int, ep_traj: bool = False):
        """(positions, symbols, charges, bond_types, bond_src, bond_dst) of one trajectory frame, fake atoms
        shown (molecule_builder.py:186-196).  Frames are token tensors: keys 'x','a','c','e' hold T frames
        (frame 0 = prior) and 'x_1_pred','a_1_pred','c_1_pred','e_1_pred' hold T-1 frames."""
        tf = self.traj_frames
        sfx = '_1_pred' if ep_traj else ''
        return extract_moldata(tf['x' + sfx][frame_idx], tf['a' + sfx][frame_idx].long(), tf['c' + sfx][frame_idx].long(),
                               tf['e' + sfx][frame_idx].long(), self.atom_type_map_in, self.fake_atoms, self.n_bond_types,
                               show_fake_atoms=True)

    def traj_mol_blocks(self, ep_traj: bool = False, align: bool = True) -> List[str]:
        """One V2000 mol block per trajectory frame, fake atoms shown as Sn and masked atoms as Se, positions
        rigidly aligned to the final frame (reference process_traj_frames, molecule_builder.py:156-214)."""
        # All frames at once in numpy (one batched SVD for the alignment, list conversions per frame): a 500-step trajectory is 501
        # frames per molecule, and per-frame torch operators on 3x3 / n-element tensors cost more in dispatch (and, on a many-core
        # host, in intra-op thread start-up) than in arithmetic.  Same results as frame_moldata() + rigid_alignment() per frame.
        tf = self.traj_frames
        sfx = '_1_pred' if ep_traj else ''
        X = tf['x' + sfx].detach().cpu().numpy().astype(np.float32, copy=True)
        A, Cq, E = (tf[k + sfx].detach().cpu().numpy().astype(np.int64) for k in 'ace')
        if align:
            X = rigid_alignment_frames(X, X[-1])
        amap = list(self.atom_type_map_in) + (['Sn'] if self.fake_atoms else []) + ['Se']      # fake atoms shown as Sn, masked atoms as Se
        src, dst = (t.numpy() for t in pair_indices(int(A.shape[1])))
        blocks = []
        for f in range(X.shape[0]):
            bt = E[f].copy()
            bt[bt == self.n_bond_types] = 0
            ok = bt != 0
            blocks.append(mol_block(X[f], [amap[i] for i in A[f]], Cq[f] - 2, src[ok], dst[ok], bt[ok], name=f'frame {f}'))
        return blocks

    def to_record(self) -> dict:
        """RDKit-free form of the molecule (what build_molecule, molecule_builder.py:268-300, consumes): plain Python / numpy."""
        return {'positions': self.positions.numpy().copy(), 'atom_types': list(self.atom_types), 'atom_charges': self.atom_charges.tolist(),
                'bond_types': self.bond_types.tolist(), 'bond_src_idxs': self.bond_src_idxs.tolist(), 'bond_dst_idxs': self.bond_dst_idxs.tolist()}

    def to_sdf_block(self) -> str:
        """V2000 mol block without RDKit (atoms, formal charges, bond orders)."""
        return mol_block(self.positions, self.atom_types, self.atom_charges, self.bond_src_idxs, self.bond_dst_idxs, self.bond_types)


def rigid_alignment(x_0: torch.Tensor, x_1: torch.Tensor) -> torch.Tensor:
    """Kabsch alignment of x_0 onto x_1 (same algorithm as reference flowmol/data_processing/priors.py:128-169):
    centre both, R from the SVD of the covariance, move x_0 into x_1's frame."""
    assert x_0.shape == x_1.shape
    m0 = x_0.mean(dim=0, keepdim=True)
    m1 = x_1.mean(dim=0, keepdim=True)
    a, b = x_0 - m0, x_1 - m1
    U, S, V = torch.svd(a.T.mm(b))
    R = V.mm(U.T)
    t = m1 - R.mm(m0.T).T
    return a.mm(R.T) + m0 + t


def rigid_alignment_frames(frames: np.ndarray, x_1: np.ndarray) -> np.ndarray:
    """rigid_alignment() of every frame (T, n, 3) onto x_1 (n, 3) with one batched SVD (float32 like the per-frame version)."""
    m0 = frames.mean(axis=1, keepdims=True)
    m1 = x_1.mean(axis=0, keepdims=True)
    a, b = frames - m0, x_1 - m1
    U, _, Vh = np.linalg.svd(np.einsum('tni,nj->tij', a, b))
    R = np.matmul(np.transpose(Vh, (0, 2, 1)), np.transpose(U, (0, 2, 1)))             # V U^T
    t = m1[None] - np.transpose(np.matmul(R, np.transpose(m0, (0, 2, 1))), (0, 2, 1))
    return (np.matmul(a, np.transpose(R, (0, 2, 1))) + m0 + t).astype(np.float32)


def build_rdkit_mol(positions, atom_types, atom_charges, bond_src, bond_dst, bond_types):
    try:
        from rdkit import Chem
        from rdkit.Geometry import Point3D
    except Exception:
        return None
    bmap = [None, Chem.rdchem.BondType.SINGLE, Chem.rdchem.BondType.DOUBLE, Chem.rdchem.BondType.TRIPLE, Chem.rdchem.BondType.AROMATIC]
    mol = Chem.RWMol()
    for sym, ch in zip(atom_types, atom_charges):
        at = Chem.Atom(sym)
        if int(ch) != 0:
            at.SetFormalCharge(int(ch))
        mol.AddAtom(at)
    for bt, s, d in zip(bond_types, bond_src, bond_dst):
        mol.AddBond(int(s), int(d), bmap[int(bt)])
    try:
        mol = mol.GetMol()
    except Exception:
        return None
    conf = Chem.Conformer(mol.GetNumAtoms())
    for i in range(mol.GetNumAtoms()):
        x, y, z = (float(v) for v in positions[i])
        conf.SetAtomPosition(i, Point3D(x, y, z))
    mol.AddConformer(conf)
    return mol


_CHG = {3: 1, 2: 2, 1: 3, -1: 5, -2: 6, -3: 7}


def mol_block(positions, atom_types, atom_charges, bond_src, bond_dst, bond_types, name: str = 'flowmol_amd') -> str:
    """MDL V2000 mol block as RDKit's writer lays it out (the reference writes SDF through Chem.SDWriter, test.py:212-257):
    counts line, atom lines with the legacy charge column, bond lines, `M  CHG` property lines (8 entries per line) for the
    charged atoms, `M  END`.  ('$$$$' is the caller's job.)"""
    def as_list(t):          # one bulk conversion instead of a tensor index per element (trajectory dumps write hundreds of frames per molecule)
        return t.tolist() if hasattr(t, 'tolist') else list(t)
    pos, chgs = as_list(positions), as_list(atom_charges)
    b_src, b_dst, b_typ = as_list(bond_src), as_list(bond_dst), as_list(bond_types)
    na, nb = len(atom_types), len(b_typ)
    lines = [name, '  flowmol_amd          3D', '', f'{na:3d}{nb:3d}  0  0  0  0  0  0  0  0999 V2000']
    charged = []
    for i, sym in enumerate(atom_types):
        x, y, z = pos[i]
        q = int(chgs[i])
        chg = _CHG.get(q, 0)
        if q != 0:
            charged.append((i + 1, q))
        lines.append(f'{x:10.4f}{y:10.4f}{z:10.4f} {sym:<3s} 0{chg:3d}  0  0  0  0  0  0  0  0  0  0')
    for k in range(nb):
        lines.append(f'{int(b_src[k]) + 1:3d}{int(b_dst[k]) + 1:3d}{int(b_typ[k]):3d}  0')
    for o in range(0, len(charged), 8):
        part = charged[o:o + 8]
        lines.append(f'M  CHG{len(part):3d}' + ''.join(f' {a:3d} {q:3d}' for a, q in part))
    lines.append('M  END')
    return '\n'.join(lines) + '\n'
