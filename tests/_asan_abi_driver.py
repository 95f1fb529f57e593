"""Run by tests/test_cpu_host.py in a child process with the clang ASan runtime LD_PRELOADed: loads the host-sanitised build of the
C-ABI library (make -C efa_xray_amd/csrc asan) and walks every entry point's argument checks -- null context, null out-pointers,
unknown options -- with no GPU call behind them.  Any AddressSanitizer / UBSan report aborts the process (halt_on_error)."""
import ctypes
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the ctypes signature table without importing the package (whose __init__ imports torch: not wanted under a preloaded sanitizer)
spec = importlib.util.spec_from_file_location("_efa_lib_table", os.path.join(ROOT, "efa_xray_amd", "_lib.py"))
src = open(spec.origin).read()
head = src[:src.index("def load_library")]
ns = {"__file__": spec.origin, "__name__": "_efa_lib_table"}
exec(compile(head, spec.origin, "exec"), ns)      # our own file: constants and SIGNATURES only
SIGNATURES = ns["SIGNATURES"]

lib = ctypes.CDLL(sys.argv[1])
for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
assert lib.efa_abi_version() == 1

n = ctypes.c_int(-1)
assert lib.efa_device_count(ctypes.byref(n)) == 0 and n.value >= 0
assert lib.efa_device_count(None) < 0
ctx = ctypes.c_void_p()
rc = lib.efa_ctx_create(0, ctypes.byref(ctx))
if n.value == 0:
    assert rc == ns["EFA_ERR_NO_DEVICE"] and not ctx.value, rc
    assert b"no CPU fallback" in lib.efa_last_error()
elif rc == 0:
    lib.efa_ctx_destroy(ctx)
assert lib.efa_ctx_create(0, None) < 0
assert lib.efa_ctx_destroy(None) == 0

called = 0
for name, (res, args) in SIGNATURES.items():
    if name in ("efa_abi_version", "efa_last_error", "efa_device_count", "efa_ctx_create", "efa_ctx_destroy", "efa_comm_unique_id"):
        continue
    fn = getattr(lib, name)
    # null context (first argument of every remaining entry), zeros / NULLs elsewhere: must be refused before anything is read
    vals = []
    for a in args:
        if a in (ctypes.c_int, ctypes.c_long, ctypes.c_size_t, ctypes.c_uint64):
            vals.append(0)
        elif a is ctypes.c_double:
            vals.append(0.0)
        else:
            vals.append(None)
    rc = fn(*vals)
    assert rc < 0, (name, rc)
    msg = lib.efa_last_error()
    assert msg and b"null context" in msg, (name, msg)
    called += 1
print("asan-abi-ok %d entries refused a null context" % called)
