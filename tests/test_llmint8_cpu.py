"""Properties of the LLM.int8 restatement (oracle/llmint8_oracle.py; Dettmers et al. 2022, as bitsandbytes applies it behind
the reference's ``load_in_8bit=True``, demo.py:27-29): CPU only."""
import numpy as np

from oracle import int8_oracle as io
from oracle import llmint8_oracle as lo
from oracle.llama_oracle import LlamaOracle, OracleConfig
from promptcache_amd.model.config import SHAPES
from promptcache_amd.model.weights import make_weights_np


def _xw(T=7, K=96, N=40, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((T, K)).astype(np.float32)
    w = (0.05 * rng.standard_normal((N, K))).astype(np.float32)
    return x, w


def test_activation_quantiser_vectorwise_and_outlier_rules():
    x, _ = _xw()
    x[2, 5] = 9.0            # one outlier entry: column 5 becomes an outlier column for EVERY row
    x[4, 11] = -6.0          # |x| == threshold counts (>=)
    ca, sca, cols, x16 = lo.quantize_activations(x)
    assert cols.tolist() == [5, 11] and not ca[:, 5].any() and not ca[:, 11].any()
    # the row statistics ignore the outlier ENTRIES only (row 2's absmax is taken without the 9.0) ...
    rest = np.abs(x16.astype(np.float32)[2, [k for k in range(96) if k != 5]]).max()
    assert sca[2] == rest and sca[2] < 6.0
    # ... every other entry is round-half-even(x * 127 / absmax_row), full range used
    keep = [k for k in range(96) if k not in (5, 11)]
    want = np.rint(x16.astype(np.float32)[:, keep] * (np.float32(127.0) / sca)[:, None]).astype(np.int8)
    assert np.array_equal(ca[:, keep], want) and (np.abs(ca).max(axis=1) == 127).all()
    # zero rows do not divide by zero
    z = np.zeros((2, 16), np.float32)
    ca0, s0, c0, _ = lo.quantize_activations(z)
    assert not ca0.any() and (s0 == 0).all() and c0.size == 0


def test_linear_decomposition_matches_the_definition():
    x, w = _xw(seed=1)
    x[:, 3] *= 20.0                                    # a systematic outlier feature, as in real LLM activations
    cb, scale = io.quantize_rows_int8(w)
    y = lo.linear(x, cb, scale)
    x16 = x.astype(np.float16).astype(np.float32)
    wdq = cb.astype(np.float32) * scale[:, None]
    # (a) the outlier column is carried in fp16 against the dequantised weights: removing it from both sides leaves the int8 part
    _, _, cols, _ = lo.quantize_activations(x)
    assert 3 in cols.tolist()
    x_in = x.copy(); x_in[:, cols] = 0
    y_in = lo.linear(x_in, cb, scale)
    fp16_part = x16[:, cols] @ (wdq[:, cols].astype(np.float16).astype(np.float32)).T
    np.testing.assert_allclose(y - y_in, fp16_part, rtol=0, atol=2e-6)
    # (b) accuracy: against the fp32 product over the DEQUANTISED weights the only error is the 8-bit activation grid
    ref = x16 @ wdq.T
    err = np.abs(y - ref).max()
    assert err < 0.03 * np.abs(ref).max() and err > 0
    # (c) threshold 0 = plain vector-wise int8 (no decomposition)
    y0 = lo.linear(x, cb, scale, threshold=0.0)
    ca, sca, cols0, _ = lo.quantize_activations(x, 0.0)
    assert cols0.size == 0
    np.testing.assert_allclose(y0, (ca.astype(np.int64) @ cb.T.astype(np.int64)).astype(np.float32) *
                               (sca[:, None] * (scale * 127)[None, :]) * lo.INV_127SQ, rtol=1e-7)
    # the decomposition is what keeps the outlier feature from wrecking the other 95: with it the error is several times smaller
    assert np.abs(y0 - ref).max() > 3 * err


def test_int8_model_oracle_tracks_the_fp32_oracle_and_uses_the_int8_weights():
    shape = SHAPES["tiny"]
    w16 = make_weights_np(shape, 5, 4.0)
    w32 = {k: v.astype(np.float32) for k, v in w16.items()}
    cfg = OracleConfig(vocab_size=shape.vocab_size, hidden_size=shape.hidden_size, intermediate_size=shape.intermediate_size,
                       num_hidden_layers=shape.num_hidden_layers, num_attention_heads=shape.num_attention_heads,
                       num_key_value_heads=shape.num_key_value_heads, rms_norm_eps=shape.rms_norm_eps, rope_theta=shape.rope_theta)
    ids = np.array([[5, 9, 200, 31, 7, 77, 400, 12]])
    pos = np.arange(8)[None]
    full, _ = LlamaOracle(cfg, w32).forward(ids, pos)
    deq, _ = LlamaOracle(cfg, io.dequantized_llama_weights(w32)).forward(ids, pos)
    i8 = lo.LlamaInt8Oracle(cfg, w32)
    assert len(i8.q) == 7 * shape.num_hidden_layers and "lm_head" not in i8.q
    got, present = i8.forward(ids, pos)
    # closer to the weight-dequantised model (same weights, 8-bit activations on top) than that one is to the fp32 model
    d_act = np.abs(got - deq).max()
    d_w = np.abs(deq - full).max()
    assert 0 < d_act < 0.5 and d_w > 0
    assert len(present) == shape.num_hidden_layers
