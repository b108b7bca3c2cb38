"""Name -> class lookup used by the model registry (protocol of ``howl/utils/class_registry.py:6-19``:
``class Foo(Base, name="foo")`` registers, ``Base.find_registered_class("foo")`` / ``Base.registered_names()`` query;
an unknown name is a ``KeyError``).  Each registry root keeps its own table in ``registered_map``."""
from typing import Dict, List


class ClassRegistry:
    registered_map: Dict[str, type] = {}

    @classmethod
    def __init_subclass__(cls, name=None, **kwargs):
        super().__init_subclass__(**kwargs)
        if name is None:
            return                      # an abstract intermediate (e.g. the registry root itself)
        table = cls.registered_map      # resolved through the MRO: the nearest root that defined a table
        table[name] = cls

    @classmethod
    def find_registered_class(cls, name: str):
        table = cls.registered_map
        if name not in table:
            raise KeyError(name)
        return table[name]

    @classmethod
    def registered_names(cls) -> List[str]:
        return [key for key in cls.registered_map]
