"""MI355X-native viscoelastic FDTD engine behind BabelBrain's solver interface.

`PropagationModel` is the drop-in for `BabelViscoFDTD.PropagationModel.PropagationModel`
(BabelIntegrationBASE.py:17-19, 43); `harness` builds caller-side inputs; `slab` runs one
domain across several GPUs (Z-slabs, neighbour halo exchange over RCCL)."""
from .PropagationModel import PropagationModel, COMPUTING_BACKEND_HIP  # noqa: F401


def ListDevices():
    """Device names, like the `ListDevices()` of the reference solver's backend modules (SelFiles/SelFiles.py:245-263)."""
    from . import _engine
    return [name for _, name in _engine.list_devices()]
