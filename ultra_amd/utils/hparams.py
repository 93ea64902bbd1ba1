"""Typed ``name=value`` hyper-parameter strings (the plugin API's configuration grammar).

Own implementation of the grammar the reference accepts (reference: ultra/utils/hparams.py,
``PARAM_RE`` :17-24, ``parse_values`` :160-259, ``HParams`` :262-574), pinned clause by clause
by tests/golden/hparams_goldens.json (generated from the reference module itself).  Pure
Python: it configures kernels, it is not on the device path.
"""
import json
import numbers
import re

# one clause:  name[ index ]? = ( scalar | [ list ] )  followed by end-of-string or a comma
_CLAUSE = re.compile(
    r'(?P<name>[a-zA-Z]\w*)(\[\s*(?P<index>\d+)\s*\])?\s*=\s*'
    r'((?P<val>[^,\[]*)|\[(?P<vals>[^\]]*)\])($|,\s*)')


def _to_bool(text):
    if text in ('true', 'True'):
        return True
    if text in ('false', 'False'):
        return False
    return bool(int(text))  # ValueError for anything that is not an integer literal


def _coerce(name, kind, value):
    """Reject lossy conversions (str<-number, bool<->number, int<-float) before casting."""
    if kind is type(None):
        return value
    bad = ((issubclass(kind, (str, bytes)) and not isinstance(value, (str, bytes)))
           or (issubclass(kind, bool) != isinstance(value, bool))
           or (issubclass(kind, numbers.Integral) and not isinstance(value, numbers.Integral))
           or (issubclass(kind, numbers.Number) and not isinstance(value, numbers.Number)))
    if bad:
        raise ValueError("Could not cast hparam '%s' of type '%s' from value %r" % (name, kind, value))
    return kind(value)


class HParams(object):
    """Named, typed hyper-parameters; types are inferred from the defaults."""

    def __init__(self, **defaults):
        self._kinds = {}
        for name, value in defaults.items():
            self.add_hparam(name, value)

    def add_hparam(self, name, value):
        if getattr(self, name, None) is not None:
            raise ValueError('Hyperparameter name is reserved: %s' % name)
        if isinstance(value, (list, tuple)):
            if not value:
                raise ValueError('Multi-valued hyperparameters cannot be empty: %s' % name)
            self._kinds[name] = (type(value[0]), True)
        else:
            self._kinds[name] = (type(value), False)
        setattr(self, name, value)

    def set_hparam(self, name, value):
        kind, is_list = self._kinds[name]
        if isinstance(value, list):
            if not is_list:
                raise ValueError('Must not pass a list for single-valued parameter: %s' % name)
            setattr(self, name, [_coerce(name, kind, v) for v in value])
        else:
            if is_list:
                raise ValueError('Must pass a list for multi-valued parameter: %s.' % name)
            setattr(self, name, _coerce(name, kind, value))

    def parse(self, text, ignore_unknown_hyperparameters=True):
        """Override values from ``"a=1,b=[1, 2],c=true"``; unknown names are printed and skipped."""
        found = {}
        pos = 0
        while pos < len(text):
            m = _CLAUSE.match(text, pos)
            if not m:
                raise ValueError('Malformed hyperparameter value: %s' % text[pos:])
            pos = m.end()
            name, index = m.group('name'), m.group('index')
            if name not in self._kinds:
                if not ignore_unknown_hyperparameters:
                    raise ValueError('Unknown hyperparameter type for %s' % name)
                print('Unknown hyperparameter type for %s' % name)
                continue
            kind = self._kinds[name][0]
            convert = _to_bool if kind is bool else kind
            scalar, items = m.group('val'), m.group('vals')

            def fail(raw):
                raise ValueError("Could not parse hparam '%s' of type '%s' with value '%s' in %s"
                                 % (name, kind.__name__, raw, text))
            if scalar is not None:
                try:
                    parsed = convert(scalar)
                except ValueError:
                    fail(scalar)
                if index is None:
                    if name in found:
                        raise ValueError("Multiple assignments to variable '%s' in %s" % (name, text))
                    found[name] = parsed
                else:
                    slot = found.setdefault(name, {})
                    if not isinstance(slot, dict) or int(index) in slot:
                        raise ValueError("Multiple assignments to variable '%s' in %s" % (name, text))
                    slot[int(index)] = parsed
            elif items is not None:
                if index is not None:
                    raise ValueError('Assignment of a list to a list index.')
                if name in found:
                    raise ValueError("Multiple assignments to variable '%s' in %s" % (name, text))
                try:
                    found[name] = [convert(e) for e in re.split('[ ,]', items) if e]
                except ValueError:
                    fail(items)
            else:
                fail('')
        return self.override_from_dict(found)

    def override_from_dict(self, values):
        for name, value in values.items():
            self.set_hparam(name, value)
        return self

    def parse_json(self, text):
        return self.override_from_dict(json.loads(text))

    def to_json(self, indent=None, separators=None, sort_keys=False):
        return json.dumps(self.values(), indent=indent, separators=separators, sort_keys=sort_keys)

    def values(self):
        return {n: getattr(self, n) for n in self._kinds}

    def get(self, key, default=None):
        return getattr(self, key) if key in self._kinds else default

    def __contains__(self, key):
        return key in self._kinds

    def __str__(self):
        return str(sorted(self.values().items()))

    def __repr__(self):
        return '%s(%s)' % (type(self).__name__, self)
