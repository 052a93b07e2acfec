"""MI355X-native hot path of nsu-ai-team/conv1d-text-vae behind the reference's estimator surface.

Same public names as the reference package (conv1d_text_vae/__init__.py:3)."""
from .tokenizer import BaseTokenizer, DefaultTokenizer
from .estimator import Conv1dTextVAE, SequenceForSeq2Seq, SequenceForVAE
from .embeddings import FastTextKeyedVectors

__all__ = ['Conv1dTextVAE', 'BaseTokenizer', 'DefaultTokenizer', 'SequenceForVAE', 'SequenceForSeq2Seq',
           'FastTextKeyedVectors']
