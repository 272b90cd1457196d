"""`from flash_attn.flash_attn_interface import ...` (reference tests/test_both_seqlens.py:3) -> the comparator."""
from . import flash_attn_func, flash_attn_varlen_func, flash_attn_with_kvcache  # noqa: F401
