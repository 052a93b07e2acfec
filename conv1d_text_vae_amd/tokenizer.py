"""Tokenizers with the reference's interface (conv1d_text_vae.py:28-138).

`DefaultTokenizer` in the reference wraps nltk's `NISTTokenizer.international_tokenize`; nltk is not available
offline, so the four substitution rules of that method (NIST mteval-v14 "international" tokenisation: separate ASCII
runs, split punctuation that is not between digits, split symbols) are restated here on top of `unicodedata`.
The known-answer tests of the reference (tests/test_conv1d_text_vae.py:17-57) pin the behaviour.
"""
from __future__ import annotations

import copy
import re
import sys
import unicodedata
from typing import List, Tuple
from xml.sax.saxutils import unescape


class BaseTokenizer:
    """Interface of the reference's tokenizers (:28-43): word spans as (begin, end) character offsets."""

    def tokenize_into_words(self, src: str) -> List[Tuple[int, int]]:
        raise NotImplementedError

    @staticmethod
    def tokenize_into_characters(src: str, bounds_of_words: List[Tuple[int, int]]) -> List[str]:
        """The characters of the words, with ONE separator for whatever lies between two words (and before the first /
        after the last): a line feed if that stretch contains one, otherwise a blank."""
        def separator(begin: int, end: int) -> List[str]:
            if end <= begin:
                return []
            return ['\n'] if '\n' in src[begin:end] else [' ']

        chars: List[str] = []
        cursor = 0
        for begin, end in bounds_of_words:
            chars.extend(separator(cursor, begin))
            chars.extend(src[begin:end])
            cursor = end
        chars.extend(separator(cursor, len(src)))
        return chars


_CLASSES = {}


def _char_class(prefix: str) -> str:
    """Regex character class of every code point whose Unicode general category starts with `prefix`."""
    if prefix not in _CLASSES:
        chars = [chr(c) for c in range(sys.maxunicode + 1) if unicodedata.category(chr(c)).startswith(prefix)]
        _CLASSES[prefix] = ''.join(re.escape(c) for c in chars)
    return _CLASSES[prefix]


class _InternationalTokenizer:
    """Restatement of nltk.tokenize.nist.NISTTokenizer.international_tokenize (return_str=False, lowercase=False)."""

    def __init__(self):
        number, punct, symbol = _char_class('N'), _char_class('P'), _char_class('S')
        self._rules = [
            (re.compile('([\x00-\x7f]+)'), r' \1 '),                                  # separate out ASCII runs
            (re.compile('([^{0}])([{1}])'.format(number, punct)), '\\1 \\2 '),         # punctuation not preceded by a digit
            (re.compile('([{1}])([^{0}])'.format(number, punct)), ' \\1 \\2'),         # punctuation not followed by a digit
            (re.compile('([{0}])'.format(symbol)), ' \\1 '),                           # symbols
        ]

    def international_tokenize(self, text: str) -> List[str]:
        text = str(text)
        text = text.replace('<skipped>', '').replace(' ', ' ')
        text = unescape(text, {'&apos;': "'", '&quot;': '"'})
        for regexp, substitution in self._rules:
            text = regexp.sub(substitution, text)
        return text.split()


class DefaultTokenizer(BaseTokenizer):
    def __init__(self, special_symbols: set = None):
        super().__init__()
        self.special_symbols = special_symbols
        self.tokenizer = _InternationalTokenizer()
        self._compile()

    def _compile(self):
        if (self.special_symbols is not None) and (len(self.special_symbols) > 0):
            re_expr = '(' + '|'.join([re.escape(cur) for cur in self.special_symbols]) + ')'
            self.re_for_special_symbols = re.compile(re_expr)
        else:
            self.re_for_special_symbols = None

    def tokenize_into_words(self, src: str) -> List[Tuple[int, int]]:
        """Spans of the NIST tokens of `src`; every occurrence of a special symbol is a token of its own and the text
        between two occurrences is tokenised separately (:57-99)."""
        if not src.strip():
            return []
        if self.re_for_special_symbols is None:
            return self._token_spans(src, 0, len(src))
        spans: List[Tuple[int, int]] = []
        cursor = 0
        for hit in self.re_for_special_symbols.finditer(src):
            spans += self._token_spans(src, cursor, hit.start())
            spans.append((hit.start(), hit.end()))
            cursor = hit.end()
        return spans + self._token_spans(src, cursor, len(src))

    def _token_spans(self, src: str, begin: int, end: int) -> List[Tuple[int, int]]:
        """Where the tokens of src[begin:end] sit in `src`: the tokenizer returns strings, so each one is looked up from
        the end of the previous one onwards (:101-114)."""
        piece = src[begin:end]
        spans: List[Tuple[int, int]] = []
        cursor = 0
        for token in self.tokenizer.international_tokenize(piece):
            token = token.strip()
            if not token:
                continue
            at = piece.find(token, cursor)
            if at < 0:
                raise ValueError('Token `{0}` cannot be found in the text `{1}`!'.format(token, piece))
            cursor = at + len(token)
            spans.append((begin + at, begin + cursor))
        return spans

    def __getstate__(self):
        return {'special_symbols': self.special_symbols}

    def __setstate__(self, state):
        self.special_symbols = state['special_symbols']
        self.tokenizer = _InternationalTokenizer()
        self._compile()

    def __copy__(self):
        # the reference's __copy__/__deepcopy__ (:128-138) build the copy but forget to return it; returning it is the
        # evident intent (get_params(deep=True) deep-copies the tokenizer, :509)
        return DefaultTokenizer(self.special_symbols)

    def __deepcopy__(self, memodict={}):
        return DefaultTokenizer(copy.deepcopy(self.special_symbols))
