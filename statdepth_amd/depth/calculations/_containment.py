"""Containment selection (reference: statdepth/depth/calculations/_containment.py:19-43,178-203).

The built-in definitions ('r2', 'simplex') are not Python functions here: they name
HIP kernels.  A user-supplied callable keeps the reference's plug-in protocol
`containment(data=<T x j DataFrame>, curve=<Series>, relax=<bool>) -> float`
(docs/index.md:124-148) and is evaluated by the generic enumerator in _functional.py,
on the host, because arbitrary Python cannot run on the GPU.
"""
from inspect import signature

BUILTIN = ('r2', 'r2_enum', 'simplex')


def _is_valid_containment(containment):
    # a string that reaches this point is not a known definition (:34-35)
    if isinstance(containment, str):
        raise ValueError(f'containment argument \'{containment}\' is invalid. Use one of '
                         f'[\'r2\', \'r2_enum\', \'simplex \'] or a pass a custom containment function.')
    params = signature(containment).parameters
    if len(params) != 3:   # only the arity is enforced (:37-41)
        raise ValueError('Custom containment method has incorrect number of parameters. '
                         'Expected 3, recieved {}'.format(len(params)))
    return containment


def _select_containment(containment):
    """Returns the built-in's name, or the validated callable (:194-203)."""
    if isinstance(containment, str) and containment in BUILTIN:
        return containment
    return _is_valid_containment(containment=containment)
