"""mini_flash_attention — FlashAttention-2 forward for AMD MI355X (gfx950 / CDNA4).

Same public surface as w4096/mini-flash-attention (reference mini_flash_attention/__init__.py:5-15).
The compute path is the hand-written HIP library ``libmfa_hip.so`` behind the ``_C`` extension; there is
no CPU or eager fallback: importing without the built extension raises.
"""

__version__ = "0.1.0"

from mini_flash_attention.interface import (
    flash_attn_func,
    flash_attn_varlen_func,
    flash_attn_with_kvcache,
)

__all__ = [
    "flash_attn_func",
    "flash_attn_varlen_func",
    "flash_attn_with_kvcache",
]
