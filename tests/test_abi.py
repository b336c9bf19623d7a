"""CPU: the C-ABI library builds, loads and exports every symbol include/npb.h declares; schema
sizes agree between header, library, oracle and the Python parser.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "nuclear_sim_amd", "libnpb.so")


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "nuclear_sim_amd", "csrc"), "-s"])
    return LIB


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "npb.h")).read()
    return sorted(set(re.findall(r"NPB_API[^;]*?\b(npb_\w+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for s in ("npb_create", "npb_destroy", "npb_step", "npb_reset", "npb_observe", "npb_get_field", "npb_set_field",
              "npb_state_bytes", "npb_last_error", "npb_version"):
        assert s in syms


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    for s in declared_symbols():
        assert hasattr(lib, s), "libnpb.so does not export %s" % s


def test_schema_sizes_agree(built_lib, oracle_lib):
    from nuclear_sim_amd.schema import SCHEMA
    lib = ctypes.CDLL(built_lib)
    assert lib.npb_num_f64() == SCHEMA.total_f64
    assert lib.npb_num_i32() == SCHEMA.total_i32
    lib.npb_state_bytes.restype = ctypes.c_size_t
    assert lib.npb_state_bytes() == SCHEMA.state_bytes()
    L = oracle_lib.lib()
    assert L.npo_num_f64() == SCHEMA.total_f64 and L.npo_num_i32() == SCHEMA.total_i32


def test_params_struct_layout_matches(built_lib):
    from nuclear_sim_amd import _lib
    from nuclear_sim_amd.schema import PARAMS
    p = _lib.default_params()
    for name, dflt, _path in PARAMS:
        assert getattr(p, name) == dflt, name
    assert p.dt == 1.0 and p.heat_source == 0 and p.mode == 0


def test_create_fails_loudly_without_a_gpu(built_lib):
    """On a CPU-only box the product path must raise, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nuclear_sim_amd import _lib
    from nuclear_sim_amd.env import BatchedPlantEnv
    with pytest.raises(_lib.NpbError):
        BatchedPlantEnv(4)
    h = ctypes.c_void_p()
    rc = _lib.load().npb_create(ctypes.byref(_lib.default_params()), 4, 0, ctypes.byref(h))
    assert rc != 0 and not h.value


def test_product_does_not_touch_the_oracle():
    """Nothing under nuclear_sim_amd/ may import, include or link oracle/."""
    pkg = os.path.join(ROOT, "nuclear_sim_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), (dirpath, f)


def test_create_rejects_arena_beyond_32bit_offsets():
    """npb_create refuses, before touching any device, a handle whose fp64 arena would pass 4 GiB
    (the kernel's column offsets are 32-bit)."""
    import ctypes
    from nuclear_sim_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    rc = L.npb_create(None, 1_100_000, 0, ctypes.byref(h))
    assert rc != 0 and not h.value
    L.npb_last_error.restype = ctypes.c_char_p
    assert b"4 GiB" in L.npb_last_error(None)


def test_create_storage_validates_its_argument():
    """npb_create_storage: unknown storage kinds are refused before any device call; fp32 storage doubles
    the number of plants one handle may carry (4 GiB of 4-byte columns)."""
    import ctypes
    from nuclear_sim_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    assert L.npb_create_storage(None, 64, 0, 7, ctypes.byref(h)) != 0 and not h.value
    L.npb_last_error.restype = ctypes.c_char_p
    assert b"storage" in L.npb_last_error(None)
    assert L.npb_create_storage(None, 2_200_000, 0, _lib.STORAGE_F32, ctypes.byref(h)) != 0 and not h.value
    assert b"4 GiB" in L.npb_last_error(None)
    L.npb_handle_step_bytes_per_plant.restype = ctypes.c_size_t
    assert L.npb_handle_step_bytes_per_plant(None) == L.npb_step_bytes_per_plant()


def test_bench_names_the_kernel_the_launcher_picks():
    """bench.py labels its roofline with the step kernel npb_step launches for the batch size; the thresholds live in the
    launcher (npb_kernels.hip, NPB_LAUNCHER(step)) and must not drift apart."""
    import importlib.util
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "nuclear_sim_amd", "csrc", "npb_kernels.hip")).read()
    four_wave_up_to = int(re.search(r"variant = npad <= (\d+) \? 5", src).group(1))
    shared_from = int(re.search(r"#define NPB_SEGMENTED_FROM \(\(size_t\)(\d+)\)", src).group(1))
    shared_up_to = int(re.search(r"#define NPB_SEGMENTED_UP_TO \(\(size_t\)(\d+)\)", src).group(1))
    assert re.search(r"\(npad <= NPB_SEGMENTED_FROM \? 2 : \(npad <= NPB_SEGMENTED_UP_TO \? 5", src)
    nt_above = int(re.search(r"#define NPB_NT_STORE_ABOVE \(\(size_t\)(\d+)\)", src).group(1))
    wide_up_to = int(re.search(r"const bool wide = .* npad <= (\d+);", src).group(1))
    assert re.search(r"const bool wide = two_wave && variant == 2 && npad <= \d+;", src)      # variant 3 never takes the wide build
    # npb_create segments the arena from the four-wave kernel's second range on
    api = open(os.path.join(root, "nuclear_sim_amd", "csrc", "npb_api.hip")).read()
    assert re.search(r"h->seg = h->pitch > %d \? 16384 : 0;" % shared_from, api)
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    old = os.environ.pop("NPB_STEP_KERNEL", None)
    try:
        assert four_wave_up_to == wide_up_to == 32768      # both are "every wave resident at once": 2 048 waves of 256 registers / 1 024 of 512
        assert bench.step_kernel_name(64) == bench.step_kernel_name(four_wave_up_to) == "npb_step4_kernel"
        assert bench.step_kernel_name(four_wave_up_to, maintenance=True) == "npb_step4_maint_kernel"
        assert bench.step_kernel_name(four_wave_up_to + 64) == bench.step_kernel_name(shared_from) == "npb_step2_kernel"
        # the four-wave kernel again, on a segmented arena, then the one-wave kernel's streaming build
        assert bench.step_kernel_name(shared_from + 64) == bench.step_kernel_name(65536) == bench.step_kernel_name(shared_up_to) == "npb_step4_kernel"
        assert shared_up_to > nt_above
        assert bench.step_kernel_name(shared_up_to + 64) == "npb_step_nt_kernel" and bench.step_kernel_name(shared_up_to + 64, "f32") == "npb_step_kernel"
        assert bench.step_kernel_name(2 * nt_above, "f32") == "npb_step_kernel" and bench.step_kernel_name(2 * nt_above + 64, "f32") == "npb_step_nt_kernel"
        # forced variants (npb_set_step_kernel / NPB_STEP_KERNEL): 1 and 4 at any size, 2 = the wide build only while it fits,
        # 3 = the 256-register build at ANY size (the round-2 launcher folded 3 into 2 before deciding `wide`)
        for n in (64, wide_up_to, wide_up_to + 64, 65536, 131072):
            assert bench.step_kernel_name(n, forced="1") == "npb_step_kernel"
            assert bench.step_kernel_name(n, forced="4") == "npb_step_nt_kernel"
            assert bench.step_kernel_name(n, forced="3") == "npb_step2_kernel"
            assert bench.step_kernel_name(n, forced="2") == ("npb_step2_wide_kernel" if n <= wide_up_to else "npb_step2_kernel")
            assert bench.step_kernel_name(n, forced="5") == "npb_step4_kernel"
        os.environ["NPB_STEP_KERNEL"] = "3"
        assert bench.step_kernel_name(64) == "npb_step2_kernel"
    finally:
        os.environ.pop("NPB_STEP_KERNEL", None)
        if old is not None:
            os.environ["NPB_STEP_KERNEL"] = old
