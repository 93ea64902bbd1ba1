"""Dotted-name class lookup used by the experiment JSONs (reference: ultra/utils/sys_tools.py:7-22)."""
import importlib


def find_class(class_str):
    """``"pkg.module.Class"`` -> class.  ImportError when the module or the class is missing."""
    module_name, _, attr = class_str.rpartition('.')
    module = importlib.import_module(module_name)
    if not hasattr(module, attr):
        raise ImportError('Class %s cannot be found in %s' % (attr, module_name))
    return getattr(module, attr)


def create_object(class_str, *args, **kwargs):
    return find_class(class_str)(*args, **kwargs)
