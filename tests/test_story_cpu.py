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


def test_add_subtitle_matches_reference_rendering():
    """``add_subtitle`` (host-side PIL overlay of the NN.jpg files) against the reference's own function where the
    reference tree is present (build container), and against its literal geometry everywhere: canvas 80 px taller,
    picture untouched, mid-string split into two white lines at x = 10."""
    import importlib.util
    import os
    import numpy as np
    from PIL import Image
    from src.inference.gen_george import add_subtitle
    img = Image.fromarray((np.arange(120 * 200 * 3) % 251).astype(np.uint8).reshape(120, 200, 3))
    text = "George climbed the tree and waved at the man with the yellow hat."
    out = add_subtitle(img, text)
    assert out.size == (200, 200) and out.mode == "RGB"
    a = np.asarray(out)
    assert np.array_equal(a[:120], np.asarray(img))                       # the picture is pasted unchanged
    bar = a[120:]
    assert bar[:33 - 1].max() == 0 and bar[:, :10].max() == 0             # nothing above the first line / left of x = 10
    assert bar[33:33 + 28].max() >= 250                                    # two lines of (anti-aliased) white text, 14 px apart
    ref_py = "/root/reference/src/inference/gen_george.py"
    if os.path.exists(ref_py):          # the reference's function, lifted out of its script (module-level code loads models)
        src = open(ref_py).read()
        start = src.index("def add_subtitle(")
        end = src.index("\nfor j in range(len(image_paths))", start)
        ns = {}
        exec("from PIL import Image, ImageDraw, ImageFont\n" + src[start:end], ns)
        ref = ns["add_subtitle"](img, text)
        assert np.array_equal(np.asarray(ref), a)
