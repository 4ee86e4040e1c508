"""Sampler registry with the reference's interface (reference sampler/base.py)."""
import warnings

__all__ = ["register_sampler", "get_sampler"]

__SAMPLER__ = {}


def register_sampler(name: str):
    def wrapper(cls):
        prev = __SAMPLER__.get(name, None)
        if prev and prev != cls:
            warnings.warn(f"Name {name} is already registered!", UserWarning)
        __SAMPLER__[name] = cls
        cls.name = name
        return cls
    return wrapper


def get_sampler(name: str, **kwargs):
    if __SAMPLER__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined.")
    return __SAMPLER__[name](**kwargs)
