"""Run-time half of pytorch_sound/data/meta/__init__.py: ``MetaType`` and the ``MetaFrame`` protocol the data set
classes read (columns / process_columns / iloc / sr).  The offline half - building the frames from a corpus on disk
(duration scans, text clean-up, the per-corpus ``*Meta.make_meta``) - is data-set bookkeeping outside the per-step path
(out of scope); frames saved by the reference (``save_meta`` json files) load unchanged.
"""
import abc
import enum
import os
from typing import List, Tuple

import pandas as pd


class MetaType(enum.Enum):
    """column kinds of a meta frame (data/meta/__init__.py:17-22)"""
    AUDIO = 1
    SCALAR = 2
    MIDI = 3
    TEXT = 4
    META = 5


_LOADED = (MetaType.AUDIO, MetaType.SCALAR, MetaType.MIDI, MetaType.TEXT)


class MetaFrame:
    """What a data set is made of (data/meta/__init__.py:25-84): subclasses give ``columns`` [(MetaType, name)], the
    pandas frame ``meta`` and the sampling rate ``sr``."""

    @property
    def process_columns(self) -> List[Tuple[MetaType, str]]:
        """the columns a data set loads: everything but MetaType.META"""
        return [c for c in self.columns if c[0] in _LOADED]

    @property
    @abc.abstractmethod
    def columns(self) -> List[Tuple[MetaType, str]]:
        raise NotImplementedError('You must define columns !')

    @property
    def column_names(self) -> List[str]:
        return [name for _, name in self.columns]

    @property
    @abc.abstractmethod
    def meta(self) -> pd.DataFrame:
        raise NotImplementedError('You must define make DataFrame!')

    @abc.abstractmethod
    def make_meta(self, *args, **kwargs):
        raise NotImplementedError('You must define make DataFrame and save it !')

    @property
    def iloc(self):
        return self.meta.iloc

    def __len__(self) -> int:
        return len(self.meta)

    @staticmethod
    def save_meta(frame_file_names: List[str], meta_path: str, all_frame: pd.DataFrame, train_frame: pd.DataFrame,
                  val_frame: pd.DataFrame):
        """[all, train, val] frames as json under ``meta_path`` (data/meta/__init__.py:113-133)"""
        assert not os.path.exists(meta_path) or os.path.isdir(meta_path)
        os.makedirs(meta_path, exist_ok=True)
        for name, frame in zip(frame_file_names, (all_frame, train_frame, val_frame)):
            frame.to_json(os.path.join(meta_path, name))
