"""A stand-in for the T5 tokenizer / encoder pair the reference pipeline holds (``ltxv.py:186-192``): the text encoder
is outside the hot path, but ``LTXMultiScalePipeline.__call__`` takes string prompts and calls
``video_pipeline.encode_prompt`` (pipeline_ltx_video.py:1833-1852), so replaying ``ltxv.py:420-445`` needs SOMETHING with
the HF call protocol.  Deterministic, tiny, and used identically on both sides: by oracle/gen/make_golden.py (``g15``)
under the REFERENCE's encode_prompt and by the tests under the product's."""
import types

import torch
from torch import nn


class FakeTokenizer:
    """UTF-8 bytes as token ids (2 + byte % 250), EOS = 1, pad = 0; the ``transformers`` call protocol the
    reference's encode_prompt uses (:376-395, :431-439)."""

    def __call__(self, text, padding=None, max_length=None, truncation=False, add_special_tokens=True,
                 return_tensors="pt", return_attention_mask=True):
        assert return_tensors == "pt"
        if isinstance(text, str):
            text = [text]
        rows = [[2 + b % 250 for b in t.encode("utf-8")] + ([1] if add_special_tokens else []) for t in text]
        if truncation and max_length is not None:
            rows = [r[:max_length] for r in rows]
        width = max_length if padding == "max_length" else max(len(r) for r in rows)
        ids = torch.zeros(len(rows), width, dtype=torch.long)
        mask = torch.zeros(len(rows), width, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r, dtype=torch.long)
            mask[i, :len(r)] = 1
        return types.SimpleNamespace(input_ids=ids, attention_mask=mask)

    def batch_decode(self, ids):
        return ["".join(chr(max(int(t) - 2, 32)) for t in row if int(t) > 1) for row in ids]


class FakeTextEncoder(nn.Module):
    """Embedding + a position term + one Linear: (input_ids, attention_mask=) -> (hidden [B, L, dim],)."""

    def __init__(self, dim, seed=0, vocab=256, max_len=256):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.emb = nn.Embedding(vocab, dim)
        self.pos = nn.Parameter(torch.zeros(max_len, dim))
        self.mix = nn.Linear(dim, dim)
        with torch.no_grad():
            self.emb.weight.copy_(torch.randn(vocab, dim, generator=g))
            self.pos.copy_(0.3 * torch.randn(max_len, dim, generator=g))
            self.mix.weight.copy_(torch.randn(dim, dim, generator=g) / dim ** 0.5)
            self.mix.bias.zero_()

    @property
    def dtype(self):
        return self.emb.weight.dtype

    def forward(self, input_ids, attention_mask=None):
        x = self.emb(input_ids) + self.pos[: input_ids.shape[1]].to(self.emb.weight.dtype)
        x = self.mix(x)
        if attention_mask is not None:
            x = x * attention_mask.unsqueeze(-1).to(x.dtype)
        return (x,)
