"""Lightweight sample metrics without RDKit: valence stability and bond-graph connectivity, computed on the
device from the sampled tokens.

Mirrors the valence part of the reference's ``SampleAnalyzer`` (flowmol/analysis/metrics.py:44-131):
``frac_atoms_stable`` / ``frac_mols_stable_valence`` through ``check_stability`` (:333-363) with the shipped
``train_data_valencies_*.json`` tables, plus the connectivity numbers the reference derives from
``Chem.GetMolFrags`` (:172-186) — ``frac_connected``, ``avg_frag_frac``, ``avg_num_components`` — which depend
only on the bond graph.  RDKit sanitisation ("validity"), REOS/ring statistics, energies and PoseBusters are
outside the hot path and not provided.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import torch

from .molecule import SampledMolecule

DATA = Path(__file__).resolve().parent / 'data' / 'valencies.json'


def load_valency_table(dataset: str = 'geom_full_kekulized'):
    """(table, explicit_aromaticity) of the dataset's training-set valencies; table[atom][charge] = list of valid
    valencies (kekulized) or of (n_aromatic, valency) pairs (explicit aromaticity).  Datasets without a shipped
    table (qm9, geom) raise FileNotFoundError like the reference does (metrics.py:67-69)."""
    allt = json.loads(DATA.read_text())
    if dataset not in allt:
        raise FileNotFoundError(f'No valency file found for dataset {dataset!r} (shipped: {sorted(allt)})')
    ent = allt[dataset]
    table = {atom: {int(ch): v for ch, v in chs.items()} for atom, chs in ent['table'].items()}
    return table, bool(ent['explicit_aromaticity'])


def encode_valency_table(table: Dict[str, Dict[int, list]], atom_type_map: Sequence[str], n_charges: int = 6,
                         explicit_aromaticity: bool = False) -> torch.Tensor:
    """(len(atom_type_map), n_charges) int32 bit masks for fm_stability: row = atom token, column = charge token
    (charge = token - 2, molecule_builder.py:245); bit v (kekulized) or bit n_arom*8+v (aromatic) = valid."""
    out = torch.zeros(len(atom_type_map), n_charges, dtype=torch.int64)
    for ai, atom in enumerate(atom_type_map):
        for ch, vals in table.get(atom, {}).items():
            ct = ch + 2
            if not 0 <= ct < n_charges:
                continue
            m = 0
            for v in vals:
                bit = (int(v[0]) * 8 + int(v[1])) if explicit_aromaticity else int(v)
                if 0 <= bit < 31:
                    m |= 1 << bit
            out[ai, ct] = m
    return out.to(torch.int32)


class SampleAnalyzer:
    """``analyze(sampled_molecules)`` -> dict with the reference's ``frac_atoms_stable`` and
    ``frac_mols_stable_valence`` keys (+ connectivity).  Needs the model the molecules came from: the counting
    runs through its engine on the GPU."""

    def __init__(self, model, dataset: Optional[str] = None):
        self.model = model
        dataset = dataset or model.cfg.n_atoms_hist
        self.valid_valency_table, self.explicit_aromaticity = load_valency_table(dataset)
        if self.explicit_aromaticity != model.explicit_aromaticity:
            raise ValueError(f'valency table of {dataset!r} and the model disagree on explicit aromaticity')
        self._table = encode_valency_table(self.valid_valency_table, model.atom_type_map, model.cfg.n_charges,
                                           self.explicit_aromaticity)

    def counts(self, tokens: Dict[str, torch.Tensor], n_atoms: torch.Tensor) -> torch.Tensor:
        """Per-molecule (B,4) int32 counts from batch tokens {'a','c','e'} (as returned by sample(return_tensors=True))."""
        eng = self.model.engine
        eng.bind(torch.as_tensor(n_atoms))
        x = tokens.get('x')
        state = eng.make_state(x if x is not None else torch.zeros(eng.N, 3), tokens['a'], tokens['c'], tokens['e'])
        fake_tok = len(self.model.atom_type_map) if self.model.fake_atoms else -1
        out = eng.stability(state, self._table, fake_tok, self.explicit_aromaticity)
        eng.synchronize()
        return out.cpu()

    def analyze(self, sampled_molecules: List[SampledMolecule]) -> Dict[str, float]:
        n_atoms = torch.tensor([int(m.x_1.shape[0]) for m in sampled_molecules])
        tokens = {k: torch.cat([getattr(m, f'{k}_1') for m in sampled_molecules]) for k in 'xace'}
        return summarize(self.counts(tokens, n_atoms))


def summarize(counts: torch.Tensor) -> Dict[str, float]:
    """Aggregate per-molecule counts like metrics.py:104-117,172-186,219-224."""
    c = counts.double()
    stable, real, comps, largest = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    nonempty = real > 0
    n_mol = counts.shape[0]
    return {
        'frac_atoms_stable': float(stable.sum() / real.sum()) if real.sum() > 0 else 0.0,
        'frac_mols_stable_valence': float((stable == real).double().sum() / n_mol),
        'frac_connected': float(((comps == 1) & nonempty).double().sum() / n_mol),
        'avg_frag_frac': float((largest[nonempty] / real[nonempty]).mean()) if nonempty.any() else 0.0,
        'avg_num_components': float(comps[nonempty].mean()) if nonempty.any() else 0.0,
    }
