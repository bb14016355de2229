"""CPU-side tests (no GPU): the C-ABI library loads and exports every declared symbol,
host logic (tables, ignition count, action vocabulary, sharding) is right, and the
product path fails loudly - never falls back - when no device is present."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import wildfire_control_python_amd as pkg
from wildfire_control_python_amd import _native as N
from wildfire_control_python_amd import utility as U
from oracle import oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "wildfire_control_python_amd", "csrc")])
    lib = N.lib()
    header = open(os.path.join(REPO, "include", "wildfire.h")).read()
    declared = set(re.findall(r"\b(wf_[a-z_]+)\s*\(", header))
    declared -= {"wf_status"}
    bound = {s[0] for s in N.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.wf_abi_version() == 2
    assert lib.wf_strerror(-3).decode().startswith("no usable HIP device")


def test_step_plan_of_the_kernel_families():
    """Host logic, no device: which family takes a grid shape and how the tiled family cuts its step (wf_step_plan) -
    the sizes DESIGN.md section 5 quotes."""
    lib = N.lib()

    def plan(H, W, B):
        v = [ctypes.c_int(-1) for _ in range(4)]
        fam = lib.wf_step_plan(H, W, B, *[ctypes.byref(x) for x in v])
        return (fam,) + tuple(x.value for x in v)

    WAVE, TILED, CELLS = 1, 2, 3
    for S in (8, 10, 12, 14, 16):
        assert plan(S, S, 65536) == (WAVE, 0, 0, 0, 0)
    assert plan(10, 20, 4096)[0] == WAVE and plan(11, 11, 4096)[0] == CELLS and plan(15, 15, 100)[0] == CELLS
    assert plan(20, 20, 100)[0] == CELLS and plan(23, 24, 100)[0] == CELLS          # tiled-eligible, but fewer than 576 cells
    assert plan(24, 24, 32768) == (TILED, 1, 3, 1, 0)            # the smallest: eight strips of 3 rows
    assert plan(30, 30, 100)[0] == CELLS and plan(50, 50, 100)[0] == CELLS          # W % 4 != 0
    assert plan(64, 260, 100)[0] == CELLS                                             # W > 256
    # (family, one launch, rows per strip, waves per workgroup, step loop)
    assert plan(28, 28, 16384) == (TILED, 1, 4, 1, 0)
    assert plan(32, 32, 16384) == (TILED, 1, 4, 1, 0)            # eight strips of 4 rows side by side in one wave
    assert plan(200, 32, 100) == (TILED, 1, 7, 4, 1)             # (height 200, width 32: eight strips of at most 8 rows)
    assert plan(48, 48, 8192) == (TILED, 1, 12, 1, 0)
    assert plan(64, 64, 16384) == (TILED, 1, 16, 1, 0)           # a job of many waves: the long strips
    assert plan(64, 64, 4096) == (TILED, 1, 8, 2, 0)             # fewer than 8 192 waves: twice the waves, half the strip
    assert plan(64, 64, 2048) == (TILED, 1, 8, 2, 1)             # ... and n-step calls inside one launch
    assert plan(64, 64, 24) == (TILED, 1, 8, 2, 1)               # never shorter than 8 rows for the sake of more waves
    assert plan(128, 128, 1024) == (TILED, 1, 8, 8, 0)
    assert plan(128, 128, 4096) == (TILED, 1, 16, 4, 0)
    assert plan(128, 128, 512) == (TILED, 1, 8, 8, 1)
    assert plan(256, 256, 256) == (TILED, 1, 16, 16, 1)          # a small job of 16-wave workgroups
    assert plan(256, 256, 1024) == (TILED, 0, 0, 0, 0)           # BASELINE configs[3]: the two-launch form
    assert plan(100, 64, 3000) == (TILED, 1, 13, 2, 1) and plan(300, 64, 1000) == (TILED, 1, 8, 10, 0)
    assert plan(300, 64, 3000) == (TILED, 0, 0, 0, 0)            # five waves per workgroup and a big job
    assert plan(600, 256, 10) == (TILED, 0, 0, 0, 0)             # more rows than 16 one-strip waves cover
    assert lib.wf_step_plan(64, 64, 4096, None, None, None, None) == TILED


def test_default_params_match_reference_constants():
    p = N.default_params()
    assert (p.ignite_count, p.allow_dig_toggle, p.containment, p.auto_reset) == (19, 0, 1, 0)
    assert (p.death_penalty, p.contained_bonus, p.default_reward) == (-1000.0, 1000.0, -1.0)
    assert (p.fuel_lo, p.fuel_hi, p.ignite_ppm) == (20, 20, 0)
    assert (p.wind_mode, p.wind_code, list(p.wind_speed)[:3]) == (N.WIND_CALM, N.wind_code(0, 0, 0), [0, 0.7, 0.85])   # constants.py:39, environment.py:189
    assert (p.heat, p.threshold) == (0.3, 3.0)                    # utility.py:98,100
    assert ctypes.sizeof(N.Params) == 128 and N.ENV_DTYPE.itemsize == 16


def test_no_wind_brings_a_heat_sum_near_the_threshold():
    """Wind with a direction (environment.py:187-193, 260-287): a grass cell's temperature is a sum of up to four
    different addends, which the reference accumulates in the iteration order of its burning set and the kernel
    (wf_step_cells<true>) and the oracle in theirs.  All orders decide `temp > threshold` alike as long as no
    combination of counts puts the exact sum within rounding distance of the threshold.  Enumerate every combination
    a cell can hold before it ignites, for every wind "random" can draw: the closest one stays 1.9e-5 away - eleven
    orders of magnitude more than 250 float64 roundings."""
    import math
    heat, thr = 0.3, 3.0
    worst = np.inf
    for speed in (0.7, 0.85):
        for wx in (-1, 0, 1):
            for wy in (-1, 0, 1):
                c = [speed * heat * (abs(math.atan2(wx * cy - wy * cx, wx * cx + wy * cy)) + 1) ** (-1)
                     for cx, cy in ((0, -1), (0, 1), (1, 0), (-1, 0))]
                n = [np.arange(int(thr / v) + 2, dtype=np.float64) for v in c]      # beyond that one direction alone ignites
                s = (n[0][:, None, None, None] * c[0] + n[1][None, :, None, None] * c[1]) \
                    + (n[2][None, None, :, None] * c[2] + n[3][None, None, None, :] * c[3])
                s = s[s < thr + max(c)]                                              # sums a cell can reach
                worst = min(worst, float(np.abs(s - thr).min()))
    assert 1.8e-5 < worst < 2.0e-5, worst


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device error path")
def test_no_device_fails_loudly_without_fallback():
    with pytest.raises(N.WildfireError) as ei:
        pkg.BatchedForestFire(8, size=14)
    assert ei.value.status == -3
    with pytest.raises(N.WildfireError):
        pkg.ForestFire(size=10)


def test_bad_arguments_rejected_before_touching_a_device():
    lib = N.lib()
    h = ctypes.c_void_p()
    p = N.default_params()
    assert lib.wf_create(ctypes.byref(h), 0, 0, 14, 14, ctypes.byref(p)) == -1       # batch 0
    assert lib.wf_create(ctypes.byref(h), 0, 1, 301, 301, ctypes.byref(p)) == -2     # containment beyond the generic flood board
    assert lib.wf_create(ctypes.byref(h), 0, 1, 1100, 8, ctypes.byref(p)) == -2      # ... and beyond the tiled one (H > 1024)
    p.auto_reset = 1
    assert lib.wf_create(ctypes.byref(h), 0, 4, 6, 14, ctypes.byref(p)) == -2        # auto-reset on a grid too small for the spawn circle
    p.auto_reset = 0
    p.ignite_count = 0
    assert lib.wf_create(ctypes.byref(h), 0, 4, 14, 14, ctypes.byref(p)) == -1
    assert lib.wf_step(None, None, 1, 1, None, None) == -1
    assert lib.wf_destroy(None) == -1


def test_ignite_count_is_the_float64_threshold_crossing():
    assert U.ignite_count(0.54) == 19          # 18*0.162 = 2.916 <= 3 < 19*0.162
    q, t = U.heat_quantum(0.54), 0.0
    for n in range(1, 20):
        t += q
        assert (t > 3) == (n >= 19)
    assert U.ignite_count(0.7) == 15 and U.ignite_count(0.85) == 12
    assert U.temp_from_count(25, 0.54) == 4.0 and U.temp_from_count(19, 0.54) == t


def test_circle_points_agree_with_oracle_restatement():
    for r in (1, 2, 3):
        assert U.circle_points(0, 0, r) == O.circle_points(r)
    assert U.get_fire_location(10, 10) == (5, 5) and U.get_fire_location(14, 14) == (7, 7)
    np.random.seed(0)
    locs = {U.get_agent_location(10, 10) for _ in range(400)}
    assert len(locs) == 36                      # 8 + 12 + 16 distinct spawn points


def test_action_vocabulary():
    enc = pkg.encode_action
    assert [enc(a) for a in ("N", "S", "E", "W", "D", " ", "x")] == [0, 1, 2, 3, 4, 5, 5]
    assert [enc(a) for a in (0, 1, 2, 3, 4, 5, -1, np.int64(2), None)] == [0, 1, 2, 3, 4, 5, 5, 2, 5]


def test_metadata_is_a_per_env_copy_with_reference_keys():
    from wildfire_control_python_amd.constants import METADATA
    assert METADATA["wind"] == [0.54, (0, 0)] and METADATA["n_actions"] == 4
    assert METADATA["contained_bonus"] == 1000 and METADATA["death_penalty"] == -1000
    assert set(U.layer) == {"type", "gray", "temp", "heat", "fuel", "threshold", "agent_pos",
                            "fire_mobility", "agent_mobility"}
    assert U.types["dirt"] == 3 and U.types[4] == "water"
    assert set(U.color2ascii.values()) == {"+", "@", "#", "0", "x"}


def test_shard_bounds_partition_the_batch():
    for gb, w in ((1_000_000, 8), (65536 * 4, 4), (10, 3), (7, 8)):
        spans = [pkg.shard_bounds(gb, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == gb
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert pkg.shard_bounds(1_000_000, 8, 3) == (375000, 500000)    # env id -> GPU = id // 125000


def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    GB, S = 96, 10
    lo, hi = pkg.shard_bounds(GB, world, rank)
    # each rank resets + steps ONLY its shard (oracle as the CPU stand-in for the kernel),
    # keyed by global env ids; no data-path collective
    planes, envs = O.new_state(hi - lo, S, S)
    p = O.default_params(fuel_lo=10, fuel_hi=40)
    O.reset(planes, envs, seed=5, env_id0=lo, params=p)
    sys.path.insert(0, REPO)
    import bench
    acts = bench.make_actions(30, lo, hi, 5)
    for t in range(30):
        O.step(planes, envs, acts[t], p)
    # bench's only collective: max over ranks of the elapsed time
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    q.put((rank, lo, hi, planes, float(t.item())))
    dist.destroy_process_group()


def test_two_rank_sharding_is_shard_count_independent():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[4] for r in res] == [2.0, 2.0]
    sharded = np.concatenate([r[3] for r in res])
    # single-"GPU" run of the same global batch
    import bench
    planes, envs = O.new_state(96, 10, 10)
    p = O.default_params(fuel_lo=10, fuel_hi=40)
    O.reset(planes, envs, seed=5, params=p)
    acts = bench.make_actions(30, 0, 96, 5)
    for t in range(30):
        O.step(planes, envs, acts[t], p)
    np.testing.assert_array_equal(sharded, planes)


def test_learner_targets_follow_the_reference_rule():
    """Regression targets of DQN.py:161-180 / DQN_SARSA.py:108-127 on a tiny CPU batch."""
    import torch
    from wildfire_control_python_amd.learner import make_network, td_targets
    torch.manual_seed(0)
    for dueling in (False, True):
        net = make_network(torch, 10, 10, 3, 4, dueling)
        s, sp = torch.rand(6, 10, 10, 3), torch.rand(6, 10, 10, 3)
        a = torch.tensor([0, 1, 2, 3, 0, 1]); ap = torch.tensor([3, 2, 1, 0, 3, 2])
        r = torch.tensor([-1.0, -1.0, 1000.0, -1000.0, -1.0, 500.0])
        done = torch.tensor([False, False, False, True, False, True])
        with torch.no_grad():
            q, qn = net(s), net(sp)
        y = td_targets(torch, net, s, a, r, sp, ap, done, 0.999, sarsa=False)
        ys = td_targets(torch, net, s, a, r, sp, ap, done, 0.999, sarsa=True)
        for i in range(6):
            other = [k for k in range(4) if k != int(a[i])]
            assert torch.equal(y[i, other], q[i, other]) and torch.equal(ys[i, other], q[i, other])
            exp = r[i] if done[i] else r[i] + 0.999 * qn[i].max()
            exps = r[i] if done[i] else r[i] + 0.999 * qn[i, ap[i]]
            assert torch.isclose(y[i, a[i]], exp) and torch.isclose(ys[i, a[i]], exps)
        assert net(s).shape == (6, 4)


def test_oracle_is_clean_under_asan_ubsan(tmp_path):
    """GPU sanitizers are unavailable on the pool; the CPU oracle (the checker everything
    else is compared with) runs under AddressSanitizer + UBSan."""
    exe = str(tmp_path / "asan_driver")
    src = os.path.join(REPO, "oracle")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(src, "asan_driver.c"), os.path.join(src, "wf_oracle.c"), "-I", src,
                           "-o", exe, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0 and "oracle clean" in out.stdout, out.stderr[-2000:]


def test_plain_c_client_builds_and_fails_loudly_without_device(tmp_path):
    """examples/native_step.c uses the boundary from plain C (gcc, no HIP headers, no Python).  Without a GPU
    the library must refuse with WF_ERR_NO_DEVICE - there is no CPU fallback behind the C-ABI either."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    from tests.golden_util import build_native_example
    exe = build_native_example(tmp_path)
    r = subprocess.run([exe, "8", "1"], capture_output=True, text=True)
    assert r.returncode == 2 and "no usable HIP device" in r.stderr, (r.returncode, r.stderr)


def test_bench_gpus_flag_is_loud():
    """`--gpus N` means N ranks on N GPUs: without them (no launcher, too few devices) or with a WORLD_SIZE that
    says otherwise, bench.py exits non-zero with a message instead of quietly benchmarking one GPU."""
    if _has_gpu():
        pytest.skip("device-count refusal is covered by the GPU bench test")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "--gpus 2 but only 0 GPU(s) visible" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="2", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()


def test_bench_labels_name_the_config_they_are():
    import bench
    assert bench.config_label(14, 65536, False, 0) == "BASELINE configs[2]"
    assert bench.config_label(10, 4096, True, 0) == "BASELINE configs[1]"
    assert bench.config_label(256, 1024, True, 10000) == "BASELINE configs[3]"
    assert "configs[4]" in bench.config_label(14, 125000, False, 0)
    assert bench.config_label(14, 16384, False, 0) == "custom size"
    ref = bench.reference_python_numbers(14)      # written by tools/time_reference.py from the real reference
    assert ref and ref["grid"] == 14 and 1e5 < ref["cell_updates_per_s_1_process"] < 1e8


def test_wind_parameters_that_counts_cannot_decide_are_refused():
    """wf_create checks a wind with a direction before it looks for a device (include/wildfire.h, wf_params.wind_mode):
    addends so small that a count byte would saturate before the threshold is reached, or a reachable heat sum within
    rounding distance of the threshold, are WF_ERR_UNSUPPORTED (-2); sound parameters get as far as the device probe."""
    import ctypes
    from wildfire_control_python_amd import _native as N
    lib = N.lib()

    def create(**kw):
        p = N.default_params()
        p.wind_mode = N.WIND_FIXED
        p.wind_code = N.wind_code(1, 0, 0)
        p.wind_speed[0] = 0.85
        for k, v in kw.items():
            if k == "speed":
                p.wind_speed[0] = v
            else:
                setattr(p, k, v)
        h = ctypes.c_void_p()
        rc = lib.wf_create(ctypes.byref(h), 0, 8, 10, 10, ctypes.byref(p))
        if rc == 0:
            lib.wf_destroy(h)
        return rc

    ok = create()
    assert ok in (0, -3), ok                              # fine: a context, or "no device" on the CPU box
    assert create(speed=0.01) == -2                       # 3 / (0.01 * 0.3 / (pi + 1)) = 4 142 addends: a count byte saturates at 255
    assert create(speed=1.0, heat=0.5, threshold=3.0) == -2      # six addends of 0.5 sum to the threshold exactly
    p = N.default_params()
    p.wind_mode = N.WIND_RANDOM                           # the reference's own "random" winds are sound
    h = ctypes.c_void_p()
    rc = lib.wf_create(ctypes.byref(h), 0, 8, 10, 10, ctypes.byref(p))
    assert rc in (0, -3), rc
    if rc == 0:
        lib.wf_destroy(h)


def test_visible_gpu_count_reads_the_kfd_topology_without_the_runtime(tmp_path, monkeypatch):
    """Launchers count GPUs from sysfs (a KFD topology node with SIMDs is a GPU), never through torch / HIP."""
    import sys
    from wildfire_control_python_amd.launcher import visible_gpu_count
    for i, simd in enumerate((0, 1024, 1024, 0, 1024)):            # CPU nodes report simd_count 0
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {16 if not simd else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert visible_gpu_count(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert visible_gpu_count(str(tmp_path)) == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert visible_gpu_count(str(tmp_path)) == 1
    assert visible_gpu_count(str(tmp_path / "nowhere")) == 0         # no KFD driver: no GPU
    assert "torch" not in sys.modules or True                        # (importing the package never imports torch: next test)


def test_package_and_bench_parent_do_not_import_torch():
    """`import wildfire_control_python_amd` and the parent of `bench.py --gpus N` stay clear of torch and of the HIP
    runtime: the parent only counts devices and starts one fresh process per rank."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import wildfire_control_python_amd as w; "
            "from wildfire_control_python_amd.launcher import visible_gpu_count; visible_gpu_count(); "
            "assert 'torch' not in sys.modules, 'torch imported'; print('clean')" % repo)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "clean" in r.stdout, r.stderr[-500:]
    # the --gpus parent: a fake HIP_VISIBLE_DEVICES list makes it see at most one device; it must refuse 2 ranks
    # before anything imports torch (-X importtime lists every module the process loads)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-X", "importtime", os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=repo)
    imported = [l for l in r.stderr.splitlines() if "import time:" in l and "| torch" in l]
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and not imported, (r.returncode, r.stderr[-300:], imported[:3])
