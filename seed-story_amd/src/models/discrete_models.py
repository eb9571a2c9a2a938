"""The only piece of the reference's ``src/models/discrete_models.py`` on the story path:
``DiscreteModleIdentity`` (reference :120-130; configs/discrete_model/discrete_identity.yaml) —
an identity "tokenizer" the de-tokenizer calls between the ViT features and ResamplerXLV2
(adapter_modules.py:417).  The SEED-X discrete tokenizer variants are out of scope."""
from torch import nn


class DiscreteModleIdentity(nn.Module):

    def __init__(self) -> None:
        super().__init__()
        self.model = nn.Identity()

    def forward(self, image_embeds, input_ids=None, text_attention_mask=None, text_embeds=None):
        return

    def encode_image_embeds(self, image_embeds):
        return self.model(image_embeds)
