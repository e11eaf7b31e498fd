"""ctypes binding of libneurst_hip.so (the C ABI declared in include/neurst_hip.h).

The prototypes are parsed from the header so Python and C cannot drift apart.  There
is NO fallback: if the shared library is missing or a symbol is absent, importing the
compute path raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'neurst_hip.h')
LIB_PATH = os.environ.get('NEURST_HIP_LIB') or os.path.join(_HERE, 'libneurst_hip.so')   # override: kernel A/B experiments (tools/exp)

F32, BF16 = 0, 1
MASK_NONE, MASK_KEYPAD, MASK_CAUSAL = 0, 1, 2

_CTYPES = {
    'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'int32_t': ctypes.c_int32, 'uint32_t': ctypes.c_uint32,
    'float': ctypes.c_float, 'void': None,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function prototype in the header."""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    protos = {}
    for m in re.finditer(r'(?:^|\n)\s*((?:const\s+)?\w+\s*\*?)\s*(neurst_\w+)\s*\(([^;{]*?)\)\s*;', text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if '*' in ret else _CTYPES[ret.replace('const', '').strip()]
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPES[a.replace('const', '').split()[0]])
        protos[name] = (restype, argtypes)
    return protos


class NeurstLibError(RuntimeError):
    pass


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NeurstLibError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                             f'(or `make -C neurst_amd/csrc`). There is no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in parse_header().items():
        fn = getattr(lib, name)        # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    m = re.search(r'#define\s+NEURST_ABI_VERSION\s+(\d+)', open(HEADER).read())
    want, got = int(m.group(1)), lib.neurst_abi_version()
    if got != want:
        raise NeurstLibError(f'{LIB_PATH} was built for ABI version {got}, include/neurst_hip.h declares {want}: rebuild the library')
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().neurst_last_error()
        raise NeurstLibError(f'{what} failed ({rc}): {msg.decode() if msg else ""}')
