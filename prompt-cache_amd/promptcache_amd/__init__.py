"""promptcache_amd -- MI355X-native prompt-cache prefill path.

Same public surface as the reference package (``promptcache/__init__.py:3-7``) for the hot path:
``CacheEngine``, ``GenerationEngine``, ``GenerationParameters``, ``Prompt``, ``CompactSpaces``,
``read_file``, ``Schema``; model adapters live in ``promptcache_amd.model`` (``Llama2``, ``CodeLlama``).
Everything below the adapters runs on hand-written HIP kernels through ``libpromptcache_hip.so``.
"""
from .pml import CompactSpaces, Prompt, Schema, read_file  # noqa: F401
from .cache_engine import CacheEngine  # noqa: F401
from .generation_engine import GenerationEngine, GenerationParameters  # noqa: F401
