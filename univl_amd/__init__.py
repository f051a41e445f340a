"""univl_amd -- MI355X (gfx950) native implementation of the UniVL hot path.

Public surface mirrors the reference's modules (`modules.modeling.UniVL`, `modules.optimization.BertAdam`);
all arithmetic runs in libunivl_hip.so (hand-written HIP kernels), there is no CPU / PyTorch fallback."""
from .modeling import UniVL  # noqa: F401
from .optimization import BertAdam, clip_grad_norm_  # noqa: F401
from ._lib import deterministic, set_deterministic  # noqa: F401
