"""CPU checks of the int8 weight quantiser (oracle/int8_oracle.py) and of its host-side product twin
(``_native.quantize_rows_int8`` is plain torch and runs on CPU tensors too): bit-identical, including exact ties."""
import numpy as np
import torch

from oracle import int8_oracle as io
from promptcache_amd import _native as n
from promptcache_amd.model.config import SHAPES
from promptcache_amd.model.weights import make_weights_np


def test_quantizer_properties():
    rng = np.random.default_rng(3)
    w = (0.05 * rng.standard_normal((64, 128))).astype(np.float32)
    w[5] = 0
    w[9, 4] = -0.5 * np.abs(w[9]).max()                     # exact tie at -63.5 -> -64 (half to even)
    q, s = io.quantize_rows_int8(w)
    assert q.dtype == np.int8 and s.dtype == np.float32 and s[5] == 1.0 and not q[5].any()
    assert np.abs(q).max(axis=1)[np.arange(64) != 5].min() == 127           # every non-zero row uses the full range
    assert q[9, 4] == -64
    err = np.abs(io.dequantize(q, s) - w)
    assert (err <= 0.5 * s[:, None] * (1 + 1e-3)).all()        # half a quantisation step (fp32 rounding of w * inv aside)


def test_host_quantizer_is_bit_identical_to_the_oracle_on_model_weights():
    w16 = make_weights_np(SHAPES["mid64"], 21, 2.0)
    checked = 0
    for k, v in w16.items():
        if v.ndim == 2 and k.startswith("l") and k.split(".")[-1] in io.LINEAR_KEYS:
            q, s = n.quantize_rows_int8(torch.from_numpy(v))
            qo, so = io.quantize_rows_int8(v.astype(np.float32))
            assert np.array_equal(q.numpy(), qo) and np.array_equal(s.numpy(), so), k
            checked += 1
    assert checked == 7 * SHAPES["mid64"].num_hidden_layers
    wd = io.dequantized_llama_weights({k: v.astype(np.float32) for k, v in w16.items()})
    assert np.array_equal(wd["embed"], w16["embed"].astype(np.float32)) and np.array_equal(wd["lm_head"], w16["lm_head"].astype(np.float32))
    assert not np.array_equal(wd["l1.down"], w16["l1.down"].astype(np.float32))


def test_int8_fragment_image_layout():
    q = torch.arange(-128, 128, dtype=torch.int16).repeat(16 * 2 * 64 // 256 + 1)[:32 * 64].to(torch.int8).view(32, 64)
    img = n.to_weight_frags_i8(q)
    assert img.shape == (2, 1, 4, 16, 2, 8) and img.dtype == torch.uint8
    # lane 16 g + m of tile t: row 16 t + m, features 8 g .. + 7 of k-step 0 then of k-step 1, offset binary
    t, g, m = 1, 2, 7
    assert torch.equal(img[t, 0, g, m, 0].to(torch.int16) - 128, q[16 * t + m, 8 * g:8 * g + 8].to(torch.int16))
    assert torch.equal(img[t, 0, g, m, 1].to(torch.int16) - 128, q[16 * t + m, 32 + 8 * g:32 + 8 * g + 8].to(torch.int16))
