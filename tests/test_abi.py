"""CPU checks of the boundary: the C-ABI library builds, loads and exports every symbol include/ebos_hip.h
declares; the ctypes table mirrors the header; the product fails loudly (no CPU fallback) without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ebos_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ebos_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from event_based_bos_amd import _hip
    from event_based_bos_amd.build import build_library

    build_library(verbose=False)
    return _hip.load_library()


def test_every_declared_symbol_is_exported(lib):
    from event_based_bos_amd import _hip

    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ebos_hip.h but not exported"
    assert sorted(_hip.SIGNATURES) == names, "ctypes table and header disagree"


def test_argument_counts_match_header():
    from event_based_bos_amd import _hip

    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in _hip.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", text, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), f"{name}: header has {n} parameters, ctypes table {len(args)}"


def test_host_only_entry_points(lib):
    from event_based_bos_amd import _hip

    assert lib.ebos_version() == _hip.ABI_VERSION == 2
    assert b"gfx950" in lib.ebos_build_info()
    cfgs = _hip.tiled_configs()
    assert (64, 64, 32) in cfgs and all(len(c) == 3 for c in cfgs)
    assert lib.ebos_bin_scratch_bytes(1000) >= 4000
    assert lib.ebos_cost_scratch_bytes(4) >= 64
    # argument validation happens before any HIP call: usable without a GPU
    rc = lib.ebos_splat_f32(None, None, 1.0, 0, 1e-6, 1, 10, 4, 5, 0, 0, None, None)
    assert rc == -1 and b"image is NULL" in lib.ebos_last_error()
    rc = lib.ebos_iwe_dense_tiled_f32(None, None, None, None, None, 0, None, 4, 5, 64, 64, 32, 1, 0, 0, None, None)
    assert rc == -1


def test_run_time_window_halo_code(lib):
    """EBOS_HALO_AUTO (include/ebos_hip.h): the `halo` arguments take -(max_halo + 256 q) with q = ceil(64 max |dt|), the bound
    ROUNDED UP; host-only entry points decode it to the built halo that sizes the workspace -- usable without a GPU."""
    from event_based_bos_amd import event_plan

    assert lib.ebos_halo_auto(32, 1.0) == -(32 + 256 * 64)
    assert lib.ebos_halo_auto(32, 0.5) == -(32 + 256 * 32) and lib.ebos_halo_auto(16, 2.0) == -(16 + 256 * 128)
    assert lib.ebos_halo_auto(32, 0.501) == -(32 + 256 * 33)            # never below the bound
    assert lib.ebos_halo_auto(32, 1e-9) == -(32 + 256 * 1)              # at least one unit
    assert lib.ebos_halo_auto(32, -1.0) == 32 and lib.ebos_halo_auto(0, 1.0) == 0 and lib.ebos_halo_auto(300, 1.0) == 300  # no auto
    built = lib.ebos_iwe_slab_workspace_bytes(720, 1280, 45, 80, 32, 1, 0, 0)
    assert built > 0 and lib.ebos_iwe_slab_workspace_bytes(720, 1280, 45, 80, lib.ebos_halo_auto(32, 1.0), 1, 0, 0) == built
    assert lib.ebos_patch_fused_supported(45, 80, lib.ebos_halo_auto(32, 1.0), 24, 32) == lib.ebos_patch_fused_supported(45, 80, 32, 24, 32)
    # max |dt| as the host knows it: dt = (t - t_ref) / (t_max - t_min) in [-f, 1 - f] for a reference time at fraction f (src/warp.py:245-253, 283-287)
    for direction, want in (("first", 1.0), ("last", 1.0), ("middle", 0.5), (0.25, 0.75), ("before", 2.0), ("after", 2.0)):
        assert event_plan.dt_bound_for(direction, True) == want
    assert event_plan.dt_bound_for("first", False) is None               # seconds: the window's length is not known on the host
    assert event_plan._max_halo(-(32 + 256 * 64)) == 32 and event_plan._max_halo(16) == 16


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    import event_based_bos_amd as ebos

    ev = np.zeros((5, 4))
    with pytest.raises(ebos.HipUnavailableError):
        ebos.Warp((4, 5)).warp_event(ev, np.zeros((2, 4, 5)), "dense-flow")
    with pytest.raises(ebos.HipUnavailableError):
        ebos.EventImageConverter((4, 5)).create_iwe(ev)
    with pytest.raises(ebos.HipUnavailableError):
        ebos.costs.functions["image_variance"]().calculate({"iwe": torch.zeros(4, 5), "omit_boundary": False})
    with pytest.raises(ebos.HipUnavailableError):
        ebos.EventPlan.build(torch.zeros(5, 4), (4, 5))
    # argument errors that the reference raises before computing anything still come first
    with pytest.raises(ValueError):
        ebos.Warp((4, 5)).warp_event(ev, np.zeros((2, 4, 5)), "dense-flow", direction=1)
    with pytest.raises(ebos.MotionModelKeyError):
        ebos.Warp((4, 5)).warp_event(ev, np.zeros((2, 4, 5)), "affine")


def test_missing_library_fails_loudly(tmp_path):
    from event_based_bos_amd import _hip

    with pytest.raises(_hip.HipUnavailableError):
        _hip.load_library(str(tmp_path / "nope.so"))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "event_based_bos_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"


@pytest.mark.gpu
def test_c_abi_is_usable_without_python(tmp_path, lib):
    """examples/c_abi_window.cpp: a plain host program (hipcc, no torch) drives the library through include/ebos_hip.h --
    raw sensor columns -> plan -> objective -> gradient -> 40 Adam iterations of the native patch-flow solver loop, as launches and as
    one resident launch -- and checks mass conservation, a finite-difference derivative, that the loss falls and that the two forms
    of the loop agree bit for bit, itself."""
    import shutil
    import subprocess

    from event_based_bos_amd import _hip

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "c_abi_window")
    libdir = os.path.dirname(_hip.LIB_PATH)
    build = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "examples", "c_abi_window.cpp"), "-L" + libdir, "-lebos_hip",
                            "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe, "150000"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.strip().endswith("OK"), (run.stdout[-1000:], run.stderr[-1000:])
    assert "solver:" in run.stdout, run.stdout[-1000:]
    # ... and the same loop as ONE resident launch through the mailbox / status protocol: the four launches' losses bit for bit
    assert "resident:" in run.stdout and "40 of 40 losses bit-identical" in run.stdout, run.stdout[-1000:]
    print(run.stdout)


def test_header_is_plain_c(tmp_path):
    """include/ebos_hip.h is the FFI contract: it must be valid C99 (what a cgo / JNI / ctypes binding consumes), not only C++."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include "ebos_hip.h"\nint main(void) { ebos_cmax_patch_problem p; (void)p; return 0; }\n')
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                        "-I" + os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("c_name,py_name", [("ebos_cmax_patch_problem", "CmaxPatchProblem"), ("ebos_dense_job", "DenseJob"),
                                             ("ebos_slab_window", "SlabWindow"), ("ebos_cmax_2dof_problem", "Cmax2dofProblem")])
def test_problem_struct_layout_matches_the_ctypes_mirror(tmp_path, c_name, py_name):
    """The structs of the ABI as a C compiler lays them out == the ctypes.Structure the host layer fills (size and every
    field offset): a field added on one side only, or a changed order, shows here and not as a wild pointer on the GPU."""
    import ctypes
    import shutil
    import subprocess

    from event_based_bos_amd import _hip

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    mirror = getattr(_hip, py_name)
    fields = [f[0] for f in mirror._fields_]
    # the header declares exactly these names, in this order
    hdr = open(os.path.join(ROOT, "include", "ebos_hip.h")).read()
    body = hdr[hdr.index("typedef struct %s {" % c_name):hdr.index("} %s;" % c_name)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    declared = []
    for stmt in body.split("{", 1)[1].split(";"):
        names = re.findall(r"[*\s,]([A-Za-z_][A-Za-z0-9_]*)\s*(?=,|$)", stmt.strip())
        declared += names
    assert declared == fields, (declared, fields)
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ebos_hip.h"', 'int main(void) {',
             '  printf("%%zu\\n", sizeof(%s));' % c_name]
    lines += [f'  printf("%zu\\n", offsetof({c_name}, {f}));' for f in fields]
    lines += ['  return 0;', '}']
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    r = subprocess.run([gcc, "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True).stdout.split()]
    want = [ctypes.sizeof(mirror)] + [getattr(mirror, f).offset for f in fields]
    assert got == want


def test_hardware_queue_default_is_set_before_the_gpu_is_touched():
    """Importing the package asks for GPU_MAX_HW_QUEUES=16 when the process has said nothing (HIP reads the variable when its runtime
    initialises; WindowPipeline counts on a queue per window in flight) and leaves a value the process chose alone."""
    import subprocess
    import sys

    code = "import os, event_based_bos_amd as e; print(os.environ.get('GPU_MAX_HW_QUEUES'), e._hip.hw_queues())"
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True).stdout.split() == ["16", "16"]
    env["GPU_MAX_HW_QUEUES"] = "4"
    assert subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True).stdout.split() == ["4", "4"]
    # opt-out: the environment is left alone and the pipeline plans for the runtime's default four queues
    env.pop("GPU_MAX_HW_QUEUES")
    env["EBOS_NO_QUEUE_ENV"] = "1"
    assert subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True).stdout.split() == ["None", "4"]


def test_hardware_queue_count_is_not_inferred_from_torchs_lazy_flag(monkeypatch):
    """A runtime another HIP user already initialised (``torch.cuda.is_available()``, a ctypes HIP call: /dev/kfd is open,
    ``torch.cuda.is_initialized()`` still False) has read GPU_MAX_HW_QUEUES already: setting it now does nothing, so the package must
    not report 16 and must not touch the environment."""
    from event_based_bos_amd import _hip

    had = os.environ.get("GPU_MAX_HW_QUEUES")
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    saved = _hip._HW_QUEUES
    try:
        monkeypatch.setattr(_hip, "_runtime_initialised", lambda: True)
        assert _hip.configure_queues(16) == 4 and "GPU_MAX_HW_QUEUES" not in os.environ and _hip.hw_queues() == 4
        monkeypatch.setattr(_hip, "_runtime_initialised", lambda: False)
        if not torch_cuda_initialised():
            assert _hip.configure_queues(12) == 12 and os.environ["GPU_MAX_HW_QUEUES"] == "12"
    finally:
        _hip._HW_QUEUES = saved
        os.environ.pop("GPU_MAX_HW_QUEUES", None)
        if had is not None:
            os.environ["GPU_MAX_HW_QUEUES"] = had


def torch_cuda_initialised():
    import torch

    return torch.cuda.is_initialized()


def test_lean_plan_scratch_is_sized_per_event_and_monotonic(lib):
    """``ebos_plan_lean_scratch_bytes`` (a host-only query): the staged streams of the single-read build -- 2 B of pixel + up to 8 B of
    timestamp per event, chunks of equal length -- plus the chunks' bin table and partials; never less than 10 B per event, never
    shrinking with the window, zero for a bad geometry."""
    import ctypes as C

    fn = lib.ebos_plan_lean_scratch_bytes
    fn.restype, fn.argtypes = C.c_size_t, [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
    prev = 0
    for n in (1, 7, 1000, 100_000, 2_000_000, 10_000_000, 50_000_000):
        b = int(fn(n, 720, 1280, 45, 80))
        assert b >= 10 * n and b >= prev, (n, b, prev)
        assert b <= 16 * n + 64 * 1024 * 1024, (n, b)      # (+ the table: chunks x bins x 4 B -- 4 B per event where a 50 M-event window needs 8192 bins)
        prev = b
    assert int(fn(100_000, 260, 346, 32, 32)) >= 10 * 100_000
    assert int(fn(-1, 720, 1280, 45, 80)) == 0 and int(fn(10, 0, 1280, 45, 80)) == 0 and int(fn(10, 720, 1280, 0, 80)) == 0
