"""Drop-in for the reference's ``src/models_clm/generation.py``.

``AutoImageTokenGenerationProcessor`` keeps the reference constructor (tokenizer,
num_img_gen_tokens) and exposes the 66 image-token ids (``img_ids_list``, reference :14-17).
On the MI355X path the processor's rule (reference :19-31) is applied *on the device* inside the
decode graph (``ss_imgproc_argmax`` / the engine's sample kernel) — no ``.item()`` host sync per
token; calling the object on tensors runs the same device kernel for API compatibility.
"""

from seedstory import ops

BOI_TOKEN = '<img>'
EOI_TOKEN = '</img>'
IMG_TOKEN = '<img_{:05d}>'


class AutoImageTokenGenerationProcessor:

    def __init__(self, tokenizer, num_img_gen_tokens=64) -> None:
        text = ''.join([BOI_TOKEN] + [IMG_TOKEN.format(int(i)) for i in range(num_img_gen_tokens)] + [EOI_TOKEN])
        self.img_ids_list = tokenizer.encode(text, add_special_tokens=False)

    def __call__(self, input_ids, scores):
        """scores [bz, vocab] on the GPU, edited in place like the reference; returns scores."""
        for i in range(input_ids.shape[0]):
            row = scores[i]
            if not row.is_contiguous():
                raise ValueError("scores rows must be contiguous")
            ops.imgproc_argmax(row, int(input_ids[i, -1]), self.img_ids_list)
        return scores


class LogitsProcessorList(list):
    """Stand-in for transformers.LogitsProcessorList (only list semantics are needed)."""
