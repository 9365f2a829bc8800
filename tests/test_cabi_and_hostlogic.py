"""CPU-side checks: the C-ABI library loads and exports every symbol include/paac_hip.h declares, the
parameter layout agrees with the oracle, host-side mirrors behave like the reference classes, and the
product never imports the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from paac_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    hdr = open(os.path.join(ROOT, "include", "paac_hip.h")).read()
    declared = set(re.findall(r"\b(paac_[a-z0-9_]+)\s*\(", hdr)) - {"paac_ctx", "paac_graph", "paac_cfg", "paac_layout"}
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libpaac_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTED_SYMBOLS), (declared ^ set(_lib.EXPORTED_SYMBOLS))
    assert _lib.load().paac_version() >= 100


def test_param_layout_matches_oracle_order():
    from oracle import network as onet
    from paac_amd import _lib
    for arch, aid in (("NIPS", 0), ("NATURE", 1)):
        for A in (4, 6, 18):
            lay = _lib.param_layout(aid, A)
            want = onet.param_shapes(arch, A)
            assert [(t["name"], t["shape"]) for t in lay["tensors"]] == [(n, tuple(s)) for n, s in want]
            assert lay["total_unpadded"] == onet.num_params(arch, A)
            assert all(t["offset"] % 4 == 0 for t in lay["tensors"]) and lay["total"] % 4 == 0
    with pytest.raises(_lib.PaacHipError):
        _lib.param_layout(1, 1)


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from paac_amd import _lib
    cfg = _lib.Cfg(device=0, arch=1, num_actions=4, max_batch=8)
    h = ctypes.c_void_p()
    rc = _lib.load().paac_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc < 0 and _lib.load().paac_last_error()


def test_product_never_imports_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "paac_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), "%s imports the oracle" % f
    code = "import sys; import paac_amd.train; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_pools_match_oracle_restatement():
    from oracle import preprocess as opre
    from paac_amd.environment import FramePool, ObservationPool
    rs = np.random.RandomState(0)
    fp = FramePool(np.empty((2, 210, 160), dtype=np.uint8), opre.max_resize)
    op = ObservationPool(np.zeros((84, 84, 4), dtype=np.uint8))
    fo, oo = opre.FramePoolOracle(), opre.ObservationPoolOracle()
    for _ in range(7):
        fr = rs.randint(0, 256, (210, 160)).astype(np.uint8)
        fp.new_frame(fr)
        fo.new_frame(fr)
        fr = rs.randint(0, 256, (210, 160)).astype(np.uint8)
        fp.new_frame(fr)
        fo.new_frame(fr)
        op.new_observation(fp.get_processed_frame())
        oo.new_observation(fo.get_processed_frame())
        assert np.array_equal(op.get_pooled_observations(), oo.get_pooled_observations())


def test_oracle_luts_regenerate_from_pil():
    """scipy.misc.imresize(img, (84, 84), interp='nearest') (atari_emulator.py:73) is PIL's NEAREST resize: the oracle's
    row / column LUTs are what PIL does to a 210x160 image (columns 52 and 73 are where it leaves floor((x+.5)*s))."""
    Image = pytest.importorskip("PIL.Image")
    from oracle import preprocess as opre
    rows = np.repeat(np.arange(210, dtype=np.uint8)[:, None], 160, axis=1)
    cols = np.repeat(np.arange(160, dtype=np.uint8)[None, :], 210, axis=0)
    got_rows = np.asarray(Image.fromarray(rows).resize((84, 84), Image.NEAREST))[:, 0]
    got_cols = np.asarray(Image.fromarray(cols).resize((84, 84), Image.NEAREST))[0, :]
    assert np.array_equal(got_rows, opre.ROW_LUT) and np.array_equal(got_cols, opre.COL_LUT)
    assert opre.COL_LUT[52] == 99 and opre.COL_LUT[73] == 139
    img = np.random.RandomState(0).randint(0, 256, (2, 210, 160)).astype(np.uint8)
    want = np.asarray(Image.fromarray(np.amax(img, axis=0)).resize((84, 84), Image.NEAREST))
    assert np.array_equal(opre.max_resize(img), want)


def test_synthetic_spec_luts_and_stats():
    from oracle import preprocess as opre
    from paac_amd import synthetic
    assert np.array_equal(synthetic.ROW_LUT, opre.ROW_LUT) and np.array_equal(synthetic.COL_LUT, opre.COL_LUT)
    env = synthetic.SyntheticEnvironment(0, 4, seed=3, terminal_p=0.1)
    s = env.get_initial_state()
    assert s.shape == (84, 84, 4) and s.dtype == np.uint8 and s[..., :3].max() == 0
    terms, rewards = 0, []
    for i in range(2000):
        o, r, t = env.next(np.eye(4)[i % 4])
        terms += t
        rewards.append(r)
        if t:
            o = env.get_initial_state()
    assert 120 < terms < 290                      # p = 0.1
    assert set(rewards) <= {-2.0, 0.0, 1.0, 3.0}
    assert 100 < np.mean(o[..., 3]) < 155


def test_arg_parser_matches_reference_flags():
    from paac_amd import train
    a = train.get_arg_parser().parse_args([])
    want = dict(game="pong", device="/gpu:0", rom_path="./atari_roms", visualize=False, e=0.1, alpha=0.99,
                initial_lr=0.0224, lr_annealing_steps=80000000, entropy_regularisation_strength=0.02, clip_norm=3.0,
                clip_norm_type="global", gamma=0.99, max_global_steps=80000000, max_local_steps=5, arch="NIPS",
                single_life_episodes=False, emulator_counts=32, emulator_workers=8, debugging_folder="logs/",
                random_start=True)                # train.py:79-98
    for k, v in want.items():
        assert getattr(a, k) == v, k
    b = train.get_arg_parser().parse_args("-g breakout -d /gpu:1 -lr 0.01 -lra 100 -ec 64 -ew 4 -df x/ -rs false --arch NATURE".split())
    assert (b.game, b.device, b.initial_lr, b.lr_annealing_steps, b.emulator_counts, b.emulator_workers,
            b.debugging_folder, b.random_start, b.arch) == ("breakout", "/gpu:1", 0.01, 100, 64, 4, "x/", False, "NATURE")


def test_runners_host_batching():
    """Runners / EmulatorRunner (own implementation) against the oracle's step loop, in-process and with workers."""
    from paac_amd.runners import EmulatorRunner, Runners
    from paac_amd.synthetic import SyntheticEnvironment
    N, A = 4, 4
    for workers in (0, 2):
        envs = [SyntheticEnvironment(i, A, seed=1, terminal_p=0.3) for i in range(N)]
        twin = [SyntheticEnvironment(i, A, seed=1, terminal_p=0.3) for i in range(N)]
        variables = [np.asarray([e.get_initial_state() for e in envs], dtype=np.uint8), np.zeros(N, np.float32),
                     np.zeros(N, np.float32), np.zeros((N, A), np.float32)]
        for e in twin:
            e.get_initial_state()
        r = Runners(EmulatorRunner, envs, workers, variables)
        r.start()
        s, rew, over, act = r.get_shared_variables()
        rs = np.random.RandomState(0)
        for _ in range(6):
            idx = rs.randint(0, A, N)
            act[...] = np.eye(A, dtype=np.float32)[idx]
            r.update_environments()
            r.wait_updated()
            for i, e in enumerate(twin):
                o, rr, t = e.next(np.eye(A)[idx[i]])
                if t:
                    o = e.get_initial_state()
                assert np.array_equal(s[i], o) and rew[i] == rr and bool(over[i]) == t
        r.stop()
        for p in r.runners:
            p.join(timeout=5)


def test_checkpoint_keys_are_the_reference_index():
    """The saver's key for every tensor of the NIPS / A=4 layout (and both optimizer slots) == the key set and shapes
    of /root/reference/pretrained/breakout/checkpoints/-80000000.index (actor_learner.py:26-27,79-82)."""
    index = "/root/reference/pretrained/breakout/checkpoints/-80000000.index"
    if not os.path.exists(index):
        pytest.skip("reference not mounted")
    import tfproto
    from oracle import network as onet
    from paac_amd import _lib
    from paac_amd.session import checkpoint_key, tensor_of_key
    lay = _lib.param_layout(_lib.ARCH_NIPS, 4)
    ours = {}
    for t in lay["tensors"]:
        for slot in (None, "OptimizerVariables", "OptimizerVariables_1"):
            key = checkpoint_key("local_learning", t["name"], slot)
            ours[key] = tuple(t["shape"])
            assert tensor_of_key(key) == (t["name"], slot)
    assert ours == tfproto.bundle_entries(index)
    assert [t["name"] for t in lay["tensors"]] == [n for n, _ in onet.param_shapes("NIPS", 4)]


def test_saver_is_atomic_and_skips_a_torn_file(tmp_path):
    """A kill during a save must not break resume: saves go through a temporary file + rename, older files are pruned
    afterwards, and latest_checkpoint skips a truncated newest file."""
    from paac_amd.session import Saver
    store = {"local_learning_1/conv1_biases": np.arange(4, dtype=np.float32)}
    got = {}
    saver = Saver(lambda: store, got.update, max_to_keep=2)
    folder = str(tmp_path)
    for step in (10, 20, 30):
        store["local_learning_1/conv1_biases"] = np.full(4, step, dtype=np.float32)
        saver.save(None, folder, step)
    names = sorted(os.listdir(folder))
    assert names == ["-20.npz", "-30.npz"]                        # max_to_keep, no temporaries left behind
    with open(os.path.join(folder, "-40.npz"), "wb") as f:         # a torn newest file
        f.write(open(os.path.join(folder, "-30.npz"), "rb").read()[:40])
    latest = Saver.latest_checkpoint(folder)
    assert latest.endswith("-30.npz")
    saver.restore(None, latest)
    assert np.array_equal(got["local_learning_1/conv1_biases"], np.full(4, 30, dtype=np.float32))
    assert Saver.latest_checkpoint(str(tmp_path / "missing")) is None


@pytest.mark.parametrize("arch,A", [("NIPS", 4), ("NATURE", 18)])
def test_weight_init_ranges(arch, A):
    """Row a10 (networks.py:24-46,63-81): every tensor is drawn from U(-d, d), d = 1/sqrt(fan_in); a conv bias uses its
    conv's kh*kw*cin, an fc bias its fc's input count (not the bias' own size)."""
    from paac_amd import _lib, networks
    lay = _lib.param_layout(_lib.ARCH_NIPS if arch == "NIPS" else _lib.ARCH_NATURE, A)
    vals = networks.initial_values(lay, np.random.RandomState(0))
    fan = {"conv1": 8 * 8 * 4}
    if arch == "NIPS":
        fan.update(conv2=4 * 4 * 16, fc3=2592, actor_output=256, critic_output=256)
    else:
        fan.update(conv2=4 * 4 * 32, conv3=3 * 3 * 64, fc4=3136, actor_output=512, critic_output=512)
    assert len(vals) == len(lay["tensors"]) == 2 * len(fan)
    for t in lay["tensors"]:
        layer = t["name"].rsplit("_", 1)[0]
        d = 1.0 / np.sqrt(fan[layer])
        v = vals[t["name"]]
        assert v.dtype == np.float32 and v.shape == tuple(t["shape"])
        assert np.abs(v).max() <= d, t["name"]
        if v.size >= 32:                                  # the range is used, not a narrower one
            assert np.abs(v).max() > 0.8 * d and abs(v.mean()) < 0.25 * d, t["name"]
    # two draws differ (unseeded default), a seeded draw repeats
    again = networks.initial_values(lay, np.random.RandomState(0))
    assert all(np.array_equal(vals[k], again[k]) for k in vals)


def test_bench_roofline_ceiling_follows_the_recorded_instruction_mix():
    """bench.py prices a contraction family against what it ran (products per multiply recorded by the library at launch
    time): fp32 MFMA 157.3 TFLOP/s, dense bf16 2500 / products otherwise, harmonic over the bodies of a paired launch --
    the conv tower at 1,536 rows (220 TFLOP/s fp32-equivalent, above the fp32-MFMA figure) stays below 1."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    kind, bodies = bench.family_work("conv_tower", 1536, "NATURE", 4, 1686693)
    assert kind == "flop" and len(bodies) == 2
    peak = bench.mfma_ceiling(bodies, (3, 6))
    assert 416.7 < peak < 833.4 and 220.0 / peak < 1.0 < 220.0 / bench.PEAK_FP32_MFMA_TFLOPS
    assert abs(bench.mfma_ceiling([1.0], (1,)) - 157.3) < 1e-9 and abs(bench.mfma_ceiling([1.0], (6,)) - 2500.0 / 6) < 1e-9
    # a paired launch is named for what the launcher paired, whatever the batch: conv2 + conv1 FLOPs, two bodies
    kind, pair = bench.family_work("conv2_conv1_wgrad", 2560, "NATURE", 18, 1700000)
    assert kind == "flop" and len(pair) == 2 and pair[0] == 2.0 * 2560 * 81 * 512 * 64 and pair[1] == 2.0 * 2560 * 400 * 256 * 32
    assert bench.mfma_ceiling(pair, (1,)) is None             # a mix that does not match the bodies: no ceiling claimed
    assert bench.family_work("clip_rmsprop", 0, "NATURE", 4, 1000)[0] == "byte"


def test_user_architecture_specs_and_library_names():
    """networks.py:117-120 through `--user_arch`: the two spellings (filter counts of the reference trunks' layer shapes, or
    filters:size:stride per layer), what is refused before any compile starts, and that an architecture's layer shapes are
    part of its library's name (two geometries never share a file)."""
    from paac_amd import build
    assert build.parse_user_arch("32,64,64,1024") == ([(32, 8, 4), (64, 4, 2), (64, 3, 1)], 1024)
    assert build.parse_user_arch("16,32,256") == ([(16, 8, 4), (32, 4, 2)], 256)
    assert build.parse_user_arch("32:8:4,64:5:2,64:3:1,512") == ([(32, 8, 4), (64, 5, 2), (64, 3, 1)], 512)
    assert build.parse_user_arch("32:4:2,64,256") == ([(32, 4, 2), (64, 4, 2)], 256)      # spellings mix per layer
    with pytest.raises(ValueError):
        build.parse_user_arch("32:8,64,256")
    stock = build.user_arch_library([(16, 8, 4), (32, 4, 2), (32, 3, 1)], 256)
    other = build.user_arch_library([(16, 8, 4), (32, 5, 2), (32, 3, 1)], 256)
    assert stock[0].endswith("libpaac_hip_user_16_32_32_256.so") and other[0] != stock[0] and "5x2" in other[0]
    for convs, fc, err in [([(16, 8, 4)], 256, NotImplementedError),                      # one conv layer
                           ([(16, 8, 4), (32, 4, 2), (32, 3, 1), (32, 3, 1)], 256, NotImplementedError),
                           ([(24, 8, 4), (32, 4, 2)], 256, NotImplementedError),          # filters not a multiple of 16
                           ([(16, 8, 4), (32, 4, 2)], 300, NotImplementedError),          # fc width not a multiple of 256
                           ([(16, 5, 2), (32, 4, 2)], 256, NotImplementedError),          # first layer: 5 x 4 channels is no K group
                           ([(16, 8, 4), (32, 4, 2), (32, 11, 1)], 256, ValueError),      # 9 x 9 in, 11 x 11 kernel: no output
                           ([(16, 8, 0), (32, 4, 2)], 256, ValueError)]:
        with pytest.raises(err):
            build.build_user_arch(convs, fc)
