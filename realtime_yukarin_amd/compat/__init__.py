"""Drop-in import surface: put this directory on sys.path and `import yukarin`, `import become_yukarin`,
`import chainer` resolve to the MI355X-backed shims.  `install()` does that for the
current interpreter."""
import sys
from pathlib import Path

COMPAT_DIR = Path(__file__).resolve().parent


def install():
    p = str(COMPAT_DIR)
    if p not in sys.path:
        sys.path.insert(0, p)
    root = str(COMPAT_DIR.parent.parent)
    if root not in sys.path:
        sys.path.insert(0, root)
