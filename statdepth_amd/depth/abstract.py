"""Abstract base of the depth result objects (reference: statdepth/depth/abstract.py:3-18)."""
from abc import ABC, abstractmethod


class AbstractDepth(ABC):
    @abstractmethod
    def ordered(self, ascending=False):
        raise NotImplementedError

    @abstractmethod
    def deepest(self, n=1):
        raise NotImplementedError

    @abstractmethod
    def outlying(self, n=1):
        raise NotImplementedError
