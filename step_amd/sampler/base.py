"""Sampler registry with the reference's interface (reference sampler/base.py)."""
from .._registry import Registry

__all__ = ["register_sampler", "get_sampler"]

__SAMPLER__ = Registry("sampler")
register_sampler = __SAMPLER__.register
get_sampler = __SAMPLER__.create
