"""TEST INFRASTRUCTURE — a deterministic stand-in for the Llama tokenizer (none is available offline: SURVEY.md §7).

Whitespace-separated words map to ids in [3, vocab) by CRC32; `__call__(text).input_ids` prepends BOS like every Llama
tokenizer (the reference's tokenizer_*_token helpers rely on it, vcoder_llava/mm_utils.py:51-53); `batch_decode` renders ids
as `t<id>` words and skips BOS / EOS / PAD when asked.  Used on BOTH sides of the COST answers fixture: by
oracle/gen_cost_golden.py when it drives the reference's loaders, and by the tests that run vcoder_amd.eval.cost_eval."""
import zlib


class _Enc:
    def __init__(self, ids):
        self.input_ids = ids


class FakeTokenizer:
    pad_token_id, bos_token_id = 0, 1

    def __init__(self, vocab_size: int = 320, eos_token_id=2):
        """eos_token_id: an id or a list of ids (HF's GenerationConfig accepts both)"""
        self.vocab_size = int(vocab_size)
        self.eos_ids = [int(e) for e in (eos_token_id if isinstance(eos_token_id, (list, tuple)) else [eos_token_id])]
        self.eos_token_id = self.eos_ids[0] if len(self.eos_ids) == 1 else list(self.eos_ids)

    @property
    def all_special_ids(self):
        return [self.pad_token_id, self.bos_token_id] + self.eos_ids

    def word_id(self, w: str) -> int:
        return 3 + zlib.crc32(w.encode("utf-8")) % (self.vocab_size - 3)   # (independent of the EOS ids)

    def __call__(self, text, **kw):
        return _Enc([self.bos_token_id] + [self.word_id(w) for w in text.split()])

    def batch_decode(self, rows, skip_special_tokens: bool = True, **kw):
        rows = rows.tolist() if hasattr(rows, "tolist") else rows
        sp = set(self.all_special_ids) if skip_special_tokens else set()
        return [" ".join(f"t{int(t)}" for t in r if int(t) not in sp) for r in rows]
