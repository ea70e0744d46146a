"""The C-ABI libraries load and export exactly what include/watsor_hip.h declares (no GPU needed): libwatsor_hip.so the part above
the WZ_DEV_BUILD section -- and nothing else that starts with wz_ --, libwatsor_hip_dev.so all of it."""
import ctypes
import os
import re
import subprocess

from watsor_amd import _lib

HEADER = os.path.join(os.path.dirname(__file__), "..", "include", "watsor_hip.h")


def declared_symbols():
    """(product, development-only) function names of the header."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    a, b = src.index("#ifdef WZ_DEV_BUILD"), src.index("#endif /* WZ_DEV_BUILD */") if "#endif /* WZ_DEV_BUILD */" in src else None
    if b is None:                                  # (the comment was stripped with the others: the section ends at its #endif)
        b = src.index("#endif", a)
    names = lambda text: sorted(set(re.findall(r"\b(wz_[a-z_0-9]+)\s*\(", text)))      # noqa: E731
    return names(src[:a] + src[b:]), names(src[a:b])


def exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if " T wz_" in l)


def test_product_library_exports_exactly_the_documented_abi():
    product, dev_only = declared_symbols()
    assert len(product) >= 30 and len(dev_only) >= 15 and not set(product) & set(dev_only)
    assert exported(_lib.LIB_PATH) == product
    assert exported(_lib.DEV_LIB_PATH) == sorted(product + dev_only)


def test_binding_covers_header():
    product, dev_only = declared_symbols()
    assert sorted(_lib.SIGNATURES) == product
    assert sorted(_lib.DEV_SIGNATURES) == dev_only
    _lib.load()
    _lib.load(dev=True)


def test_product_library_reads_four_environment_settings():
    """WZ_LANES, WZ_STREAMS, WZ_GRAPH, WZ_SCHEDULE (+ GPU_MAX_HW_QUEUES, which it sets for the HIP runtime): every tuning knob the sources read
    through wz_dev_getenv() exists in the development build only -- its name is not even in the product binary."""
    import glob
    csrc = os.path.join(os.path.dirname(HEADER), "..", "watsor_amd", "csrc")
    text = "".join(open(f).read() for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.cpp")) +
                   glob.glob(os.path.join(csrc, "*.h")))
    read_directly = set(re.findall(r'[^_]getenv\("([A-Z0-9_]+)"\)', text))
    assert read_directly == {"WZ_GRAPH", "WZ_LANES", "WZ_STREAMS", "WZ_SCHEDULE"}, read_directly
    knobs = set(re.findall(r'"(WZ_[A-Z0-9_]+)"', text)) - read_directly
    assert len(knobs) >= 40
    product, dev = open(_lib.LIB_PATH, "rb").read(), open(_lib.DEV_LIB_PATH, "rb").read()
    assert all(k.encode() in product for k in read_directly)
    leaked = sorted(k for k in knobs if k.encode() + b"\0" in product)
    assert not leaked, leaked
    assert sum(k.encode() + b"\0" in dev for k in knobs) >= 40


def test_no_gpu_calls_fail_cleanly():
    for dev in (False, True):
        lib = _lib.load(dev=dev)
        assert lib.wz_device_count() >= 0
        h = ctypes.c_void_p()
        rc = lib.wz_create(b"/nonexistent/mi355x.bin", 0, 1, 64, 64, ctypes.byref(h))
        assert rc == _lib.WZ_ENOENT and "not found" in _lib.last_error(lib)
        try:
            _lib.check(rc, lib=lib)
            assert False
        except FileNotFoundError:
            pass


def test_schedule_is_a_process_wide_setting_fixed_on_first_use():
    """include/watsor_hip.h: wz_set_schedule / wz_get_schedule (no GPU needed).  Fresh processes: the option wins over nothing, the
    environment decides when nothing was set, and once fixed only the value in force is accepted."""
    import sys
    root = os.path.join(os.path.dirname(__file__), "..")
    script = ("import sys; sys.path.insert(0, %r)\n"
              "from watsor_amd import runtime as r\n"
              "import os\n"
              "mode = sys.argv[1]\n"
              "if mode == 'option':\n"
              "    r.set_schedule('latency'); assert r.get_schedule() == 'latency'\n"
              "    r.set_schedule('latency')\n"
              "    try:\n"
              "        r.set_schedule('throughput'); print('NO ERROR')\n"
              "    except ValueError as e:\n"
              "        print('refused:', e)\n"
              "elif mode == 'env':\n"
              "    print(r.get_schedule())\n"
              "    try:\n"
              "        r.set_schedule('bogus')\n"
              "    except ValueError as e:\n"
              "        print('bad name:', e)\n" % os.path.abspath(root))
    env = {k: v for k, v in os.environ.items() if k != "WZ_SCHEDULE"}
    p = subprocess.run([sys.executable, "-c", script, "option"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "refused:" in p.stdout and "already fixed for the latency schedule" in p.stdout, p.stdout + p.stderr
    p = subprocess.run([sys.executable, "-c", script, "env"], env=dict(env, WZ_SCHEDULE="latency"), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.splitlines()[0] == "latency" and "bad name:" in p.stdout, p.stdout + p.stderr
    p = subprocess.run([sys.executable, "-c", script, "env"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.splitlines()[0] == "throughput", p.stdout + p.stderr
