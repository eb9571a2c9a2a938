"""CPU: host-side bookkeeping of bench.py that the GPU records depend on (no device work is imported or launched)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_prompt_schedule_and_decode_groups():
    b = _bench()
    # BOS + (48 caption + 66 image tokens) per pair, at most WINDOW pairs: S = 115, 229, ..., 913, 913, ...
    assert [b.prompt_len(i) for i in (0, 1, 7, 8, 9, 24)] == [115, 229, 913, 913, 913, 913]
    assert b.T_GEN == 48 + 66 + 1 and b.WINDOW == 8 and len(b.IMG_IDS) == 66
    assert b.slot_groups(8) == [8] and b.slot_groups(4) == [4] and sum(b.slot_groups(16)) == 16 and max(b.slot_groups(16)) <= 8
    assert sum(b.slot_groups(12)) == 12 and all(1 <= g <= 8 for g in b.slot_groups(12))


def test_sink_mode_bookkeeping_matches_the_story_module():
    """The --sink path of bench.mllm_part keeps, per eviction, exactly what seedstory.story.sink_keep_indices keeps; replayed here on
    ids only for a 25-step story: 17 evictions, sink prefix 4 + 24 per eviction, and the KV rows a slot ever holds stay inside the cache
    capacity main() sizes for --sink (the numbers tests/test_zz_bench_flows_gpu.py asserts on the GPU come from this arithmetic)."""
    b = _bench()
    sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
    from seedstory.story import sink_keep_indices
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import seedstory_oracle as O
    story_len = 25
    cap = (4 + 24 * max(0, story_len - b.WINDOW) + 1 + 114 * (b.WINDOW + 1) + 128 + 127) // 128 * 128      # main(), SINK branch
    ids = [b.BOS] + [5] * b.CAPTION + b.IMG_IDS
    imgs, sink, log, peak = 1, 0, [], 0
    for step in range(1, story_len):
        ids = ids + [5] * b.CAPTION + b.IMG_IDS          # advance_context in sink mode: append, no prompt surgery
        imgs += 1
        keep = len(ids) - 65
        kv = sink + keep
        while imgs > b.WINDOW:
            bi, ei = ids.index(b.IMG_IDS[0]) + sink, ids.index(b.IMG_IDS[-1]) + sink
            idx, new_sink = sink_keep_indices(kv, bi, ei, sink, sink == 0)
            # (like the reference's torch.cat, the 3 rows behind </img> appear twice: once in the sink window, once as the head of the
            # window that slides down — vis_george_sink.py:273-287; the oracle's own index function says the same)
            assert (idx, new_sink) == O.sink_evict_indices(kv, bi, ei, sink, sink == 0) and idx[-1] == kv - 1
            log.append((len(idx), kv - len(idx)))
            ids = ids[ids.index(b.IMG_IDS[-1]) + 1:]
            imgs -= 1
            kv, sink = len(idx), new_sink
        peak = max(peak, kv + 65 + b.T_GEN)              # + the 65-row continuation + the generated tokens of the step
    assert len(log) == story_len - b.WINDOW == 17
    assert log[0] == (875, 87) and log[1] == (899, 90) and sink == 4 + 24 * 17
    assert all(d in (87, 90) for _, d in log)            # first eviction drops the BOS-side rows too, later ones 114 - 24
    assert peak <= cap == 1664
