"""MI355X-native viscoelastic FDTD engine behind BabelBrain's solver interface.

`PropagationModel` is the drop-in for `BabelViscoFDTD.PropagationModel.PropagationModel`
(BabelIntegrationBASE.py:17-19, 43); `harness` builds caller-side inputs; `slab` runs one
domain across several GPUs (Z-slabs, neighbour halo exchange over RCCL)."""
from .PropagationModel import PropagationModel, COMPUTING_BACKEND_HIP  # noqa: F401
