"""junctiontree_amd - MI355X-native sum-product belief propagation behind the API of
jluttine/junction-tree.

    import junctiontree_amd as jt
    tree = jt.create_junction_tree(factors, sizes)
    marginals = tree.propagate(values)

The public names mirror the reference's `junctiontree/__init__.py:1` (`from .junctiontree
import *` plus the sub-modules).  All message passing runs in hand-written HIP kernels for
gfx950 through the C ABI of `lib/libjtprop.so` (see include/jtprop.h); importing the
package works anywhere, but computing requires the built library and a GPU - there is no
CPU fallback.
"""

from . import computation, construction, junctiontree, sum_product  # noqa: F401
from .junctiontree import *  # noqa: F401,F403
from .junctiontree import __all__ as _jt_all

__all__ = list(_jt_all) + ["computation", "construction", "sum_product", "junctiontree"]


def __getattr__(name):
    # `__version__` is the library's (`jtp_version()`: "jtprop <version> (gfx950, ...) src:<id>"), read when asked for: importing the
    # package must not need the built library
    if name == "__version__":
        from . import _capi
        try:
            return _capi.lib().jtp_version().decode().split()[1]
        except Exception:       # noqa: BLE001 - no library in this tree (a source checkout before build())
            return "0.0.0+unbuilt"
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
