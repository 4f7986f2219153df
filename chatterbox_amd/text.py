"""Host-side text front-end (CPU string work, microseconds; out of the kernel scope -- SURVEY.md section 2).

`punc_norm` follows the behaviour of the reference helper (mtl_tts.py:71-110); `MTLTokenizer` is the generic path
of the reference tokenizer wrapper (models/tokenizers/tokenizer.py:255-305: lowercase + NFKD + `[lang]` tag +
`[SPACE]` substitution + a `tokenizers` BPE).  The optional language-specific normalisers (Cangjie, kakasi, Hangul
decomposition, Hebrew diacritics, Russian stress) depend on third-party packages and data files that are not part
of the hot path; they are not re-implemented here and a warning is emitted when such a language is requested.
"""
import logging
import unicodedata

import torch

log = logging.getLogger(__name__)

_PUNC_MAP = {"...": ", ", "…": ", ", ":": ",", " - ": ", ", ";": ", ", "—": "-", "–": "-", " ,": ",",
             "“": '"', "”": '"', "‘": "'", "’": "'"}
_ENDERS = (".", "!", "?", "-", ",", "、", "，", "。", "？", "！")
_NEEDS_EXTRA = {"zh", "ja", "he", "ko", "ru"}


def punc_norm(text: str, enders=_ENDERS) -> str:
    """Multilingual variant (mtl_tts.py:71-110): sentence enders include the CJK marks."""
    if not text:
        return "You need to add some text for me to talk."
    if text[0].islower():
        text = text[0].upper() + text[1:]
    text = " ".join(text.split())
    for old, new in _PUNC_MAP.items():
        text = text.replace(old, new)
    text = text.rstrip(" ")
    if not text.endswith(enders):
        text += "."
    return text


def punc_norm_en(text: str) -> str:
    """English ChatterboxTTS variant (tts.py:26-61): same replacement table, ASCII-only sentence enders."""
    return punc_norm(text, enders=(".", "!", "?", "-", ","))


_TURBO_PUNC_MAP = {"…": ", ", ":": ",", "—": "-", "–": "-", " ,": ",", "“": '"', "”": '"', "‘": "'", "’": "'"}


def punc_norm_turbo(text: str) -> str:
    """The Turbo/Nano variant of the helper (reference tts_turbo.py:30-66): a shorter replacement table, ASCII enders."""
    if not text:
        return "You need to add some text for me to talk."
    if text[0].islower():
        text = text[0].upper() + text[1:]
    text = " ".join(text.split())
    for old, new in _TURBO_PUNC_MAP.items():
        text = text.replace(old, new)
    text = text.rstrip(" ")
    if not text.endswith((".", "!", "?", "-", ",")):
        text += "."
    return text


class MTLTokenizer:
    SOT, EOT, SPACE = "[START]", "[STOP]", "[SPACE]"

    def __init__(self, vocab_file_path):
        from tokenizers import Tokenizer
        self.tokenizer = Tokenizer.from_file(str(vocab_file_path))
        voc = self.tokenizer.get_vocab()
        assert self.SOT in voc and self.EOT in voc, "tokenizer vocabulary lacks [START]/[STOP]"

    def encode(self, txt, language_id=None, lowercase=True, nfkd_normalize=True):
        if lowercase:
            txt = txt.lower()
        if nfkd_normalize:
            txt = unicodedata.normalize("NFKD", txt)
        if language_id in _NEEDS_EXTRA:
            log.warning("language '%s' uses an optional third-party normaliser in the reference that is not bundled; "
                        "falling back to the generic grapheme path", language_id)
        if language_id:
            txt = f"[{language_id.lower()}]{txt}"
        return self.tokenizer.encode(txt.replace(" ", self.SPACE)).ids

    def text_to_tokens(self, text, language_id=None, **kw):
        return torch.IntTensor(self.encode(text, language_id=language_id, **kw)).unsqueeze(0)


class EnTokenizer(MTLTokenizer):
    def encode(self, txt, language_id=None, **kw):
        return self.tokenizer.encode(txt.replace(" ", self.SPACE)).ids
