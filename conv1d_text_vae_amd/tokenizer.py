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
    def tokenize_into_words(self, src: str) -> List[Tuple[int, int]]:
        raise NotImplemented

    @staticmethod
    def tokenize_into_characters(src: str, bounds_of_words: List[Tuple[int, int]]) -> List[str]:
        characters_list = []
        start_pos = 0
        for bounds_of_cur_word in bounds_of_words:
            if bounds_of_cur_word[0] > start_pos:
                characters_list.append('\n' if '\n' in src[start_pos:bounds_of_cur_word[0]] else ' ')
            characters_list += list(src[bounds_of_cur_word[0]:bounds_of_cur_word[1]])
            start_pos = bounds_of_cur_word[1]
        if start_pos < len(src):
            characters_list.append('\n' if '\n' in src[start_pos:] else ' ')
        return characters_list


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
        prep = src.strip()
        if len(prep) == 0:
            return []
        if self.re_for_special_symbols is None:
            return self.__tokenize_text(src)
        bounds_of_subphrases = []
        start_pos = 0
        for search_res in self.re_for_special_symbols.finditer(src):
            if (search_res.start() < 0) or (search_res.end() < 0):
                break
            bounds_of_subphrases.append(('', (start_pos, search_res.start())))
            bounds_of_subphrases.append((src[search_res.start():search_res.end()], (search_res.start(), search_res.end())))
            start_pos = search_res.end()
        if start_pos < len(src):
            bounds_of_subphrases.append(('', (start_pos, len(src))))
        bounds_of_tokens = []
        for cur_subphrase in bounds_of_subphrases:
            if len(cur_subphrase[0]) == 0:
                text = src[cur_subphrase[1][0]:cur_subphrase[1][1]]
                for cur_token in self.__tokenize_text(text):
                    bounds_of_tokens.append((cur_subphrase[1][0] + cur_token[0], cur_subphrase[1][0] + cur_token[1]))
            else:
                bounds_of_tokens.append((cur_subphrase[1][0], cur_subphrase[1][1]))
        return bounds_of_tokens

    def __tokenize_text(self, src: str) -> List[Tuple[int, int]]:
        prep = src.strip()
        if len(prep) == 0:
            return []
        bounds_of_tokens = []
        end_pos = 0
        for cur_token in filter(lambda it2: len(it2) > 0,
                                map(lambda it1: it1.strip(), self.tokenizer.international_tokenize(src))):
            start_pos = src.find(cur_token, end_pos)
            if start_pos < 0:
                raise ValueError('Token `{0}` cannot be found in the text `{1}`!'.format(cur_token, src))
            end_pos = start_pos + len(cur_token)
            bounds_of_tokens.append((start_pos, end_pos))
        return bounds_of_tokens

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
