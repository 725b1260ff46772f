import abc


class Interface(abc.ABC):
    """encode / decode contract of pytorch_sound/interface/__init__.py:4-15."""

    @abc.abstractmethod
    def encode(self, *inputs):
        raise NotImplementedError

    @abc.abstractmethod
    def decode(self, *inputs):
        raise NotImplementedError
