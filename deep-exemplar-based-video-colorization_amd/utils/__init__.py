"""Overlay of the reference's `utils` package (/root/reference/utils/__init__.py is a REGULAR package,
so a namespace directory of the same name earlier on sys.path would simply lose to it).

With this package directory in front of the reference on sys.path, `import utils` resolves here and
`__path__` is extended with every other `utils/` directory on sys.path: `utils.util` is this
package's HIP-backed module (which forwards every name it does not define to the reference's
`utils/util.py`), while `utils.util_distortion`, `utils.warping`, ... keep resolving to the
reference's own files.  See INTEGRATION.md §1 and tests/test_dropin_boundary.py.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
