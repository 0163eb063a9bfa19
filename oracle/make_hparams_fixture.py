"""TEST INFRASTRUCTURE (build container only; reads /root/reference, never imported by the product).

Derives, mechanically, the ``hyper_parameters`` a Lightning checkpoint of each shipped model configuration would hold, and
writes them to tests/golden/hparams_from_yaml.json:

  * every YAML under /root/reference/configs (flowmol3.yml, dev.yml) and configs/configs_dataprocessing (the four
    geom_*.yaml) is read with yaml.FullLoader like ``read_config_file`` (flowmol/model_utils/load.py:7-11);
  * the keyword arguments of ``FlowMol(...)`` are assembled exactly as ``model_from_config`` assembles them
    (load.py:13-49): atom_type_map, the two processed_data_dir files, sample_interval, n_mols_to_sample,
    vector_field_config, interpolant_scheduler_config, lr_scheduler_config and ``**config['mol_fm']``;
  * Lightning's ``save_hyperparameters()`` (flowmol.py:169) stores EVERY constructor argument, so the defaults of
    ``FlowMol.__init__`` (flowmol.py:29-55) -- taken from the reference's source with ``ast`` because the module itself
    needs pytorch_lightning -- fill the arguments a YAML does not pass.

The fixture is data (keyword values); tests/test_host_logic.py feeds each entry to check_reference_hparams /
from_reference_hparams and compares the result with the preset that claims to be that YAML, and
tests/parity_util.py builds its Lightning-shaped checkpoint from it (not from the preset it is compared with).

    python -m oracle.make_hparams_fixture
"""
import ast
import json
from pathlib import Path

import yaml

REF = Path('/root/reference')
OUT = Path(__file__).resolve().parent.parent / 'tests' / 'golden' / 'hparams_from_yaml.json'


def flowmol_init_defaults():
    """{argument: default} of the reference's FlowMol.__init__, literal defaults only (flowmol/models/flowmol.py:29-55)."""
    src = (REF / 'flowmol' / 'models' / 'flowmol.py').read_text()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'FlowMol')
    init = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == '__init__')
    args = init.args.args[1:]                                   # without self
    defaults = init.args.defaults
    out, required = {}, []
    first_default = len(args) - len(defaults)
    for i, a in enumerate(args):
        if i < first_default:
            required.append(a.arg)
        else:
            out[a.arg] = ast.literal_eval(defaults[i - first_default])
    return required, out, init.lineno


def kwargs_from_config(config: dict) -> dict:
    """The keyword arguments model_from_config passes to FlowMol (load.py:13-49), paths as POSIX strings."""
    processed = Path(config['dataset']['processed_data_dir'])
    kw = dict(atom_type_map=config['dataset']['atom_map'],
              n_atoms_hist_file=str(processed / 'train_data_n_atoms_histogram.pt'),
              marginal_dists_file=str(processed / 'train_data_marginal_dists.pt'),
              sample_interval=config['training']['evaluation']['sample_interval'],
              n_mols_to_sample=config['training']['evaluation']['mols_to_sample'],
              vector_field_config=config['vector_field'],
              interpolant_scheduler_config=config['interpolant_scheduler'],
              lr_scheduler_config=config['lr_scheduler'])
    dup = set(kw) & set(config['mol_fm'])
    assert not dup, f'mol_fm repeats explicit keyword(s) {dup}: FlowMol(...) would raise'
    kw.update(config['mol_fm'])
    return kw


def main():
    required, defaults, lineno = flowmol_init_defaults()
    out = {'_source': {'reference_init': f'flowmol/models/flowmol.py:{lineno}', 'mapping': 'flowmol/model_utils/load.py:13-49',
                       'required_arguments': required, 'init_defaults': defaults}}
    files = sorted((REF / 'configs').glob('*.yml')) + sorted((REF / 'configs' / 'configs_dataprocessing').glob('*.yaml'))
    for f in files:
        config = yaml.load(f.read_text(), Loader=yaml.FullLoader)
        kw = kwargs_from_config(config)
        unknown = sorted(set(kw) - set(required) - set(defaults))
        assert not unknown, f'{f.name}: FlowMol.__init__ has no argument(s) {unknown}'
        missing = [r for r in required if r not in kw]
        assert not missing, f'{f.name}: required argument(s) {missing} not passed'
        hp = dict(defaults)
        hp.update(kw)
        out[f.name] = {'yaml': str(f.relative_to(REF)), 'passed_keywords': sorted(kw), 'hyper_parameters': hp}
    OUT.write_text(json.dumps(out, indent=1, sort_keys=True) + '\n')
    print(OUT, OUT.stat().st_size, 'bytes;', ', '.join(k for k in out if not k.startswith('_')))


if __name__ == '__main__':
    main()
