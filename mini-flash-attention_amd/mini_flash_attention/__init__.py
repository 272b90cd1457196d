"""mini_flash_attention — FlashAttention-2 forward for AMD MI355X (gfx950 / CDNA4).

Same public surface as w4096/mini-flash-attention (reference mini_flash_attention/__init__.py:5-15): three
functions and ``__version__``.  The compute path is the hand-written HIP library ``libmfa_hip.so`` behind the
``_C`` extension; there is no CPU or eager fallback: importing without the built extension raises.
"""
from . import interface as _interface

__version__ = "0.1.0"
__all__ = ["flash_attn_func", "flash_attn_varlen_func", "flash_attn_with_kvcache"]

flash_attn_func = _interface.flash_attn_func
flash_attn_varlen_func = _interface.flash_attn_varlen_func
flash_attn_with_kvcache = _interface.flash_attn_with_kvcache
