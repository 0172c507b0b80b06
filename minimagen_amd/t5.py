"""Text-encoder front-end with the reference's names (minimagen/t5.py)."""
from __future__ import annotations

MAX_LENGTH = 256
DEFAULT_T5_NAME = 't5_small'

# minimagen/t5.py:10-21
T5_VERSIONS = {
    't5_small': {'tokenizer': None, 'model': None, 'handle': 't5-small', 'dim': 512, 'size': .24},
    't5_base': {'tokenizer': None, 'model': None, 'handle': 't5-base', 'dim': 768, 'size': .890},
    't5_large': {'tokenizer': None, 'model': None, 'handle': 't5-large', 'dim': 1024, 'size': 2.75},
    't5_3b': {'tokenizer': None, 'model': None, 'handle': 't5-3b', 'dim': 1024, 'size': 10.6},
    't5_11b': {'tokenizer': None, 'model': None, 'handle': 't5-11b', 'dim': 1024, 'size': 42.1},
    'small1.1': {'tokenizer': None, 'model': None, 'handle': 'google/t5-v1_1-small', 'dim': 512, 'size': .3},
    'base1.1': {'tokenizer': None, 'model': None, 'handle': 'google/t5-v1_1-base', 'dim': 768, 'size': .99},
    'large1.1': {'tokenizer': None, 'model': None, 'handle': 'google/t5-v1_1-large', 'dim': 1024, 'size': 3.13},
    'xl1.1': {'tokenizer': None, 'model': None, 'handle': 'google/t5-v1_1-xl', 'dim': 2048, 'size': 11.4},
    'xxl1.1': {'tokenizer': None, 'model': None, 'handle': 'google/t5-v1_1-xxl', 'dim': 4096, 'size': 44.5},
}


def get_encoded_dim(name: str) -> int:
    """minimagen/t5.py:87-90"""
    return T5_VERSIONS[name]['dim']


def t5_encode_text(text, name: str = 't5_base', max_length=MAX_LENGTH):
    """minimagen/t5.py:31-84.  The HIP T5 encoder (K16) is not built yet in this round and there are no
    tokenizer/checkpoint files offline; pass ``text_embeds`` / ``text_masks`` to ``Imagen.sample`` instead."""
    raise NotImplementedError("t5_encode_text: HIP T5 encoder (SURVEY K16) not built yet; supply text_embeds/text_masks")
