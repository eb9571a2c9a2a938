"""CPU: id-level story bookkeeping (window schedule, masks, recompute eviction) and the sink index spec."""
import torch

import seedstory_oracle as O
from seedstory.story import StoryContext, sink_keep_indices


def _ctx(window=3):
    return StoryContext(bos_id=1, boi_id=900, eoi_id=965, img_placeholder_ids=list(range(901, 965)), window=window)


def test_prompt_schedule_matches_survey():
    c = _ctx(window=8)
    c.start(list(range(10, 58)), torch.zeros(1, 256, 8))
    assert len(c.ids) == 1 + 48 + 66                                  # S = 115
    for step in range(1, 8):
        c.append_step(list(range(100, 148)), torch.zeros(1, 256, 8))
        assert len(c.ids) == 115 + 114 * step
    m, em = c.masks("cpu")
    assert int(m.sum()) == 64 * 8 and em.shape == (8,)
    assert not c.over_window()
    c.append_step(list(range(100, 148)), torch.zeros(1, 256, 8))
    assert c.over_window() and c.evict_recompute() == 1
    assert c.image_embeds.shape[0] == 8 and c.ids[0] == 1 and c.ids.count(900) == 8


def test_sink_indices_equal_oracle_spec():
    for (n, b, e, s, first) in [(300, 20, 85, 0, True), (290, 52, 117, 28, False), (500, 30, 95, 52, False)]:
        assert sink_keep_indices(n, b, e, s, first) == O.sink_evict_indices(n, b, e, s, first)
