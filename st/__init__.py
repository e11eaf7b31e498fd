"""Drop-in import path: `import st.models.TransformerTPE`, `from st.models.tools.utils import dense`
etc. resolve to the SAME module objects as neurst_amd.st.* (the reference's package is called `st`)."""
import importlib
import importlib.abc
import importlib.util
import sys

import neurst_amd.st as _real


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real_name):
        self.real_name = real_name

    def create_module(self, spec):
        return importlib.import_module(self.real_name)

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith('st.'):
            real = 'neurst_amd.' + fullname
            try:
                if importlib.util.find_spec(real) is None:
                    return None
            except ModuleNotFoundError:
                return None
            return importlib.util.spec_from_loader(fullname, _AliasLoader(real))
        return None


sys.meta_path.insert(0, _AliasFinder())
__path__ = list(_real.__path__)
