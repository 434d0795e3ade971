"""MI355X-native GVD decode/train hot path.  Import as `import gvd_amd`."""
from . import opts, synth  # noqa: F401
