"""flowmol_amd: MI355X-native implementation of the FlowMol3 sampling hot path.

Drop-in surface of the reference package for that path (reference flowmol/__init__.py):

    import flowmol_amd as flowmol
    model = flowmol.load_pretrained('flowmol3').cuda().eval()
    molecules = model.sample_random_sizes(n_molecules=10, n_timesteps=250)
"""
from .config import VFConfig
from .model import FlowMol, load_pretrained, pretrained_model_names, read_checkpoint
from .molecule import SampledMolecule

__all__ = ['FlowMol', 'SampledMolecule', 'VFConfig', 'load_pretrained', 'pretrained_model_names', 'read_checkpoint']
__version__ = '0.1.0'
