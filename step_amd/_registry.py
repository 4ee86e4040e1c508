"""Name -> class registries shared by the operator and sampler packages.

Behaviour follows the reference's two registries (forward_operator/base.py:8-25,
sampler/base.py): registering a second, different class under a taken name warns and
replaces the entry, the decorated class gets a ``name`` attribute, and looking up a name
nobody registered raises NameError."""
import warnings


class Registry(dict):
    def __init__(self, kind):
        super().__init__()
        self.kind = kind

    def register(self, name):
        def decorate(cls):
            taken = self.get(name)
            if taken is not None and taken is not cls:
                warnings.warn("Name %s is already registered!" % name, UserWarning)
            self[name] = cls
            cls.name = name
            return cls
        return decorate

    def create(self, name, **kwargs):
        if name not in self:
            raise NameError("Name %s is not defined." % name)
        return self[name](**kwargs)
