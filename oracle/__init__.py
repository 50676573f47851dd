"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/llama_oracle.py header).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``prompt-cache_amd/``) never does.
"""
