"""fp16-operand floor of the oracle (test infrastructure, see oracle/__init__.py).

`with RoundGemmOperands(torch.float16): oracle.detector.forward(...)` runs the fp32 oracle with every operand of every
contraction (F.linear / matmul / bmm / einsum / conv2d: activations AND weights) rounded to fp16 on entry and everything
else -- accumulation, residual streams, normalisations, softmaxes, GEMM outputs -- left in fp32.  That is the smallest error
ANY implementation that feeds fp16 operands to MFMA can have against the fp32 reference on the same weights: it is printed
beside the product's error in the parity ladder (tests/gpu_diag.py --ladder) so that "how far from 1e-3" can be split into
"inherent to fp16 operands" and "added by this implementation".  round_outputs=True additionally rounds every contraction's
result (fp16 storage of GEMM outputs)."""
import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode

_GEMM = {F.linear, torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__, torch.Tensor.__rmatmul__, torch.bmm,
         F.conv2d, torch.einsum, torch.baddbmm}


class RoundGemmOperands(TorchFunctionMode):
    def __init__(self, dtype=torch.float16, round_outputs=False):
        super().__init__()
        self.dtype, self.round_outputs = dtype, round_outputs

    def _r(self, t):
        if torch.is_tensor(t) and t.is_floating_point():
            return t.to(self.dtype).to(t.dtype)
        if isinstance(t, (list, tuple)):
            return type(t)(self._r(x) for x in t)
        return t

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _GEMM:
            out = func(*tuple(self._r(a) for a in args), **kwargs)
            return self._r(out) if self.round_outputs else out
        return func(*args, **kwargs)
