"""SchemaCache._pack: how the encode groups its scaffold passes into right-padded batches (host logic, CPU)."""
import types

from promptcache_amd.cache_engine import SchemaCache


def _sc(batch_invariant=True):
    sc = SchemaCache.__new__(SchemaCache)
    sc.lm = types.SimpleNamespace(hf_model=types.SimpleNamespace(batch_invariant=batch_invariant))
    return sc


def _padded(groups, lengths):
    return sum(len(g) * max(lengths[i] for i in g) for g in groups)


def test_pack_groups_similar_lengths_and_is_a_partition():
    # the persona-structured schema's suffix passes (tokens up to the last owned one)
    lengths = [0, 232, 240, 226, 229, 265, 250, 258, 271, 267, 259, 270, 255, 263, 155, 149, 161, 158, 150, 256, 240, 249, 262,
               251, 174, 169, 181, 160, 177]
    mine = list(range(1, len(lengths)))
    sc = _sc()
    groups = sc._pack(mine, lengths, 1)
    assert sorted(i for g in groups for i in g) == mine
    for g in groups:
        assert [lengths[i] for i in g] == sorted((lengths[i] for i in g), reverse=True)
        assert len(g) <= sc.encode_rows_max and (len(g) == 1 or len(g) * lengths[g[0]] <= sc.encode_token_budget)
    real = sum(lengths[i] for i in mine)
    assert _padded(groups, lengths) <= 1.07 * real                  # greedy fill-to-budget padded this plan by 17 %
    # never worse than one batch per budget-full of longest-first passes, forward cost included
    order = sorted(mine, key=lambda i: -lengths[i])
    greedy, cur = [], []
    for i in order:
        if cur and (len(cur) + 1) * lengths[cur[0]] > sc.encode_token_budget:
            greedy.append(cur); cur = []
        cur.append(i)
    greedy.append(cur)
    cost = lambda gs: _padded(gs, lengths) + sc.encode_forward_cost * len(gs)     # noqa: E731
    assert cost(groups) <= cost(greedy)


def test_pack_edge_cases():
    sc = _sc()
    assert sc._pack([], [], 1) == []
    assert sc._pack([3], [0, 0, 0, 9000], 1) == [[3]]                            # a pass longer than the budget travels alone
    assert sc._pack([0, 1, 2, 3, 4], [5, 5, 5, 5, 5], 2) == [[0, 1], [2, 3], [4]]   # the reference's batch_size knob: in order
    same = sc._pack(list(range(40)), [100] * 40, 1)
    assert sorted(i for g in same for i in g) == list(range(40)) and all(len(g) <= sc.encode_rows_max for g in same)
    assert _sc(batch_invariant=False)._pack([0, 1, 2], [7, 9, 8], 1) == [[0], [1], [2]]   # LLM.int8: one scaffold per call


def test_pack_counts_the_trunk_prefix_rows_against_the_arena_byte_budget():
    # 32 short union members behind a long trunk: every row of the group arena also holds its trunk prefix, so the rows
    # x suffix-width budget alone would put all of them into one forward (32 x 8 k x 1.6 MB/token at 13b = 400 GB)
    sc = _sc()
    lengths = [0] + [40] * 32
    prefix = [0] + [8000] * 32
    mine = list(range(1, 33))
    row_bytes = 40 * 2 * 40 * 128 * 2 * 2                    # 13b: L * 2 * Hkv * D * fp16, + residual planes
    free = sc._pack(mine, lengths, 1)
    assert max(len(g) for g in free) == 32
    sc.encode_arena_bytes = 48 * 2 ** 30
    capped = sc._pack(mine, lengths, 1, prefix, row_bytes)
    assert sorted(i for g in capped for i in g) == mine
    for g in capped:
        assert len(g) == 1 or len(g) * (8000 + 40) * row_bytes <= sc.encode_arena_bytes
    assert max(len(g) for g in capped) < 32


def _plan_of(schema_xml):
    from tests import helpers as H
    from promptcache_amd.pml import Schema
    lm = H.TokOnlyLM()
    lm.hf_model = types.SimpleNamespace(batch_invariant=True)
    sc = SchemaCache.__new__(SchemaCache)
    sc.lm, sc._jobs = lm, None
    sc.schema = Schema(H.llama_formatter()(schema_xml), lm)
    return sc


def test_union_member_equal_to_or_prefixing_the_default_member_still_runs_one_row():
    #  the scaffold cut behind the last owned token removed the "at least one token runs" invariant: a member
    # whose tokens equal / prefix the default member's shared its whole kept range with the trunk -> zero-width passes
    body = " ".join(f"word{i} of the shared system text" for i in range(12))
    doc = " ".join(f"fact{i} about the document" for i in range(14))
    xml = (f'<schema name="dup"><system>{body}</system><user><union scaffold="a">'
           f'<module name="a">{doc} and a tail that only the first member has</module>'
           f'<module name="b">{doc}</module><module name="c">{doc} and a tail that only the first member has</module>'
           f'</union></user><assistant>ok</assistant></schema>')
    sc = _plan_of(xml)
    jobs, prefix = sc._plan_with_prefix()
    need = sc._need(jobs, prefix)
    assert len(jobs) >= 3 and any(p > 0 for p in prefix)
    for i in range(len(jobs)):
        assert need[i] - prefix[i] >= 1, (i, need[i], prefix[i])
    costs = sc.pass_costs()
    assert costs == [need[i] - prefix[i] for i in range(len(jobs))] and sc.plan_cost() == sum(costs)
    # the packer never sees a non-positive width
    groups = sc._pack([i for i in range(len(jobs)) if prefix[i] > 0], costs, 1)
    assert all(costs[i] > 0 for g in groups for i in g)


def test_rows_kslices_fills_the_chip_with_k_slices_for_wide_panels():
    """LlamaHIP.rows_kslices: 65..288 rows -> 256 // (N / 128) slices (wide panels), 289..512 -> half of that (two row blocks per
    panel), clamped to 1..8 up to 128 rows and to 1..6 above (measured: profiles/r04_variants.txt); other row counts keep the
    model's default."""
    import types
    from promptcache_amd.model.llama_hip import LlamaHIP
    m = types.SimpleNamespace(SKINNY_MAX_ROWS=64, kslices=4)
    f = lambda T, N: LlamaHIP.rows_kslices(m, T, N)       # noqa: E731
    assert f(259, 5120) == 6 and f(259, 4096) == 6 and f(128, 4096) == 8 and f(129, 4096) == 6 and f(100, 8192) == 4 and f(70, 64) == 8
    assert f(402, 5120) == 3 and f(512, 4096) == 4
    assert f(12, 4096) == 4 and f(64, 4096) == 4 and f(600, 4096) == 4
    assert f(259, 128 * 300) == 1
