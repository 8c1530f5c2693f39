import functools
import inspect


def change_default_args(**kwargs):
    """Returns a decorator that re-defaults constructor kwargs of a layer class
    (reference: rslo/torchplus/tools.py:47-60).  Positional arguments win over the new defaults."""

    def wrap(layer_class):
        sig = inspect.signature(layer_class.__init__)
        names = [n for n in sig.parameters][1:]

        class Defaulted(layer_class):
            def __init__(self, *args, **kw):
                for key, val in kwargs.items():
                    if key not in kw and (key not in names or names.index(key) >= len(args)):
                        kw[key] = val
                super().__init__(*args, **kw)

        Defaulted.__name__ = layer_class.__name__
        Defaulted.__qualname__ = layer_class.__qualname__
        functools.update_wrapper(Defaulted.__init__, layer_class.__init__)
        return Defaulted

    return wrap
