"""-m gpu: the HIP path (through the C ABI) against the oracle on the same seeded inputs and the SAME BVH bytes.

Bars: reference-sphere mode — float within 1e-6 (SURVEY §8c tolerance), RGBA8 within 1 LSB;
path tracer — ray count identical, image RMSE <= 1e-4 (north_star) — in practice bit-identical.
"""
import os

import numpy as np
import pytest

from test_oracle_reference_kat import check_against_kat

HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.gpu
RMSE_TOL = 1e-4  # BASELINE.json north_star: "images within 1e-4 RMSE of the CPU reference"


def rmse(a, b):
    return float(np.sqrt(np.mean((a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)) ** 2)))


def run_both(P, pto, r, sd, params, width=4, count=False):
    r.SetScene(sd, width)
    r.Params = params
    if count:
        r.Params.flags |= P.native.PT_FLAG_COUNT_VISITS
    st = r.Render(0.0)
    img = r.ReadFramebuffer()
    info = r.BvhInfo()
    osc = pto.Scene(sd, (info.width,) + r.BvhRead())
    assert osc.validate_bvh()[0] == 0
    ref, ost = pto.render(osc, params)
    return img, st, ref, ost


def assert_parity(img, st, ref, ost, exact=True):
    assert st.rays == ost.rays, (st.rays, ost.rays)
    assert st.paths == ost.paths
    e = rmse(img, ref)
    nbad = int((img != ref).any(axis=-1).sum())
    assert e <= RMSE_TOL, (e, nbad)
    assert np.array_equal(img[..., 3], ref[..., 3])
    if exact:
        assert nbad == 0, f"{nbad} pixels differ, rmse {e}"


def test_reference_sphere_full_frame(P, pto, renderer):
    """a1-a4 of SURVEY §8a: the reference's CSMain at its own 1920x1080 (App.cs:27, Renderer.cs:1020)."""
    renderer.Params = P.make_params(1920, 1080, mode=P.native.PT_REFERENCE_SPHERE)
    st = renderer.Render(0.0)
    f, b = renderer.ReadFramebuffer(), renderer.ReadFramebufferRGBA8()
    check_against_kat(f, b)                       # the reference-pinned known answers
    of, ob = pto.reference_sphere(1920, 1080)
    assert np.abs(f - of).max() <= 1e-6
    assert np.abs(b.astype(int) - ob.astype(int)).max() <= 1
    assert (f != of).sum() == 0 and (b != ob).sum() == 0, "expected bit-exact: IEEE div/sqrt on both sides"
    assert st.rays == 1920 * 1080


@pytest.mark.parametrize("w,h", [(1, 1), (63, 5), (65, 64), (100, 37)])
def test_reference_sphere_ragged_sizes(P, pto, renderer, w, h):
    renderer.Params = P.make_params(w, h, mode=P.native.PT_REFERENCE_SPHERE)
    renderer.Render(0.0)
    of, ob = pto.reference_sphere(w, h)
    assert np.array_equal(renderer.ReadFramebuffer(), of) and np.array_equal(renderer.ReadFramebufferRGBA8(), ob)


@pytest.mark.parametrize("width", [2, 4, 68, 72, 73])
def test_c1_cornell(P, pto, renderer, width):
    """BASELINE config C1: Cornell box, 4 Lambert spheres + area light, 256x256, 4 spp."""
    sd = P.make_scene(P.native.PT_SCENE_CORNELL, 0, 0x5EED0001, 256, 256)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(256, 256, spp=4, max_depth=8), width, count=True)
    assert_parity(img, st, ref, ost)
    assert (st.node_visits, st.tri_tests, st.sphere_tests) == (ost.node_visits, ost.tri_tests, ost.sphere_tests)


@pytest.mark.parametrize("width", [2, 4, 68, 72, 73])
def test_c4_glass_metal_depth16(P, pto, renderer, width):
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 0x5EED0001, 200, 150)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(200, 150, spp=8, max_depth=16), width)
    assert_parity(img, st, ref, ost)


@pytest.mark.parametrize("width", [2, 4, 68, 72, 73])
def test_c3_triangle_soup(P, pto, renderer, width):
    sd = P.make_scene(P.native.PT_SCENE_TRIANGLE_SOUP, 50000, 0x5EED0001, 160, 120)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(160, 120, spp=4, max_depth=8), width, count=True)
    assert_parity(img, st, ref, ost)
    assert (st.node_visits, st.tri_tests) == (ost.node_visits, ost.tri_tests)


def test_packed_extend_kernel_is_identical(P, pto, renderer):
    """PT_FLAG_EXTEND_PACKED (ballot/mbcnt lane refill) must not change a single bit nor a single visit count."""
    sd = P.make_scene(P.native.PT_SCENE_TRIANGLE_SOUP, 30000, 3, 200, 150)
    p = P.make_params(200, 150, spp=5, max_depth=8, flags=P.native.PT_FLAG_EXTEND_PACKED)
    img, st, ref, ost = run_both(P, pto, renderer, sd, p, 4, count=True)
    assert_parity(img, st, ref, ost)
    assert (st.node_visits, st.tri_tests, st.sphere_tests) == (ost.node_visits, ost.tri_tests, ost.sphere_tests)
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 3, 130, 70)
    p = P.make_params(130, 70, spp=3, max_depth=12, flags=P.native.PT_FLAG_EXTEND_PACKED)
    assert_parity(*run_both(P, pto, renderer, sd, p, 2))


def test_pool_extend_kernel_is_identical(P, pto, renderer):
    """PT_FLAG_EXTEND_POOL (a wavefront owns 128 queue entries, refills idle lanes while it traverses, shades the pool at full
    width) must not change a single bit nor a single visit count — on every node layout, with holes, streams and deep paths."""
    N = P.native
    for width in (2, 4, 68, 72, 73):
        sd = P.make_scene(N.PT_SCENE_TRIANGLE_SOUP, 30000, 3, 200, 150)
        p = P.make_params(200, 150, spp=5, max_depth=8, streams=3, flags=N.PT_FLAG_EXTEND_POOL)
        img, st, ref, ost = run_both(P, pto, renderer, sd, p, width, count=True)
        assert st.reserved[0] == 3
        assert_parity(img, st, ref, ost)
        assert (st.node_visits, st.tri_tests, st.sphere_tests) == (ost.node_visits, ost.tri_tests, ost.sphere_tests)
    sd = P.make_scene(N.PT_SCENE_CORNELL_GLASS, 0, 3, 130, 70)
    for streams, spp in ((1, 3), (8, 11)):
        p = P.make_params(130, 70, spp=spp, max_depth=12, streams=streams, flags=N.PT_FLAG_EXTEND_POOL)
        assert_parity(*run_both(P, pto, renderer, sd, p, 0))
    sd = P.make_scene(N.PT_SCENE_CORNELL_TESS, 60000, 7, 333, 97)
    p = P.make_params(333, 97, spp=9, max_depth=8, streams=8, flags=N.PT_FLAG_EXTEND_POOL)
    img, st, ref, ost = run_both(P, pto, renderer, sd, p, 0, count=True)
    assert_parity(img, st, ref, ost)
    assert (st.node_visits, st.tri_tests, st.sphere_tests) == (ost.node_visits, ost.tri_tests, ost.sphere_tests)


def test_c5_tessellated_cornell(P, pto, renderer):
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_TESS, 60000, 0x5EED0001, 160, 120)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(160, 120, spp=4, max_depth=8), 4)
    assert_parity(img, st, ref, ost)
    # free cross-check: tessellated walls are the same surfaces as C2's => same picture up to edge ties
    sd2 = P.make_scene(P.native.PT_SCENE_CORNELL, 0, 0x5EED0001, 160, 120)
    img2, _, _, _ = run_both(P, pto, renderer, sd2, P.make_params(160, 120, spp=4, max_depth=8), 4)
    assert np.mean(np.abs(img[..., :3] - img2[..., :3]) > 1e-3) < 0.02


def test_1m_triangles_full_hd_properties(P, pto, renderer):
    """BASELINE full size (1920x1080, 2^20 triangles): size-independent properties + oracle at reduced spp."""
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_TESS, 1 << 20, 0x5EED0001, 1920, 1080)
    renderer.SetScene(sd, 4)
    renderer.Params = P.make_params(1920, 1080, spp=2, max_depth=8)
    st = renderer.Render(0.0)
    img = renderer.ReadFramebuffer()
    assert np.isfinite(img).all() and (img[..., :3] >= 0).all()
    assert (img[..., 3] == 1.0).all()                       # every pixel finished exactly spp paths
    assert st.paths == 1920 * 1080 * 2 and st.paths <= st.rays <= st.paths * 8
    st2 = renderer.Render(0.0)                              # idempotent / deterministic
    assert st2.rays == st.rays and np.array_equal(renderer.ReadFramebuffer(), img)
    info = renderer.BvhInfo()
    osc = pto.Scene(sd, (info.width,) + renderer.BvhRead())
    assert osc.validate_bvh()[0] == 0
    ref, ost = pto.render(osc, renderer.Params)
    assert ost.rays == st.rays
    assert rmse(img, ref) <= RMSE_TOL and np.array_equal(img, ref)


def test_headline_configuration_as_benchmarked(P, pto, renderer):
    """The configuration bench.py times, exactly: 1M-triangle Cornell, 1920x1080, 64 spp, 8 sample streams, max depth 8, default
    node layout and default extend-kernel choice (probed). Frame bit-identical to the oracle, ray count equal."""
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_TESS, 1 << 20, 0x5EED0001, 1920, 1080)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(1920, 1080, spp=64, max_depth=8, streams=8), 0)
    assert renderer.BvhInfo().width == 68
    assert_parity(img, st, ref, ost)
    assert st.paths == 1920 * 1080 * 64 and (img[..., 3] == 1.0).all()
    st2 = renderer.Render(0.0)  # the second frame runs with the kernel the first one picked
    assert st2.rays == st.rays and np.array_equal(renderer.ReadFramebuffer(), ref)


def test_c2_and_c3_as_benchmarked(P, pto, renderer):
    """BASELINE configs[1] exactly (Cornell, 1920x1080, 64 spp, 8 streams) and configs[2] (1M-triangle soup, 1920x1080, 8 streams) at
    16 of its 64 spp, default layout and probed extend kernel (the soup ends up on the lane-packing kernel): frames bit-identical to
    the oracle, ray counts equal."""
    sd = P.make_scene(P.native.PT_SCENE_CORNELL, 0, 0x5EED0001, 1920, 1080)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(1920, 1080, spp=64, max_depth=8, streams=8), 0)
    assert_parity(img, st, ref, ost)
    sd = P.make_scene(P.native.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 0x5EED0001, 1920, 1080)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(1920, 1080, spp=16, max_depth=8, streams=8), 0)
    assert_parity(img, st, ref, ost)
    # 2 samples per stream: the frame is over before the in-frame probe can compare anything, so whole frames decide — the first ran
    # on the one-ray-per-lane kernel, the second runs on the lane-packing one, the third on the winner (here: lane-packing)
    # (a frame that had to allocate its buffers is not counted as a measurement, hence one or two frames on kernel 1 first)
    kernels = [int(st.reserved[0])]
    for _ in range(3):
        st2 = renderer.Render(0.0)
        kernels.append(int(st2.reserved[0]))
        assert st2.rays == ost.rays and np.array_equal(renderer.ReadFramebuffer(), ref)
    assert kernels in ([1, 2, 2, 2], [1, 1, 2, 2]), kernels


def test_c3_and_c4_at_their_exact_sizes(P, pto, renderer):
    """BASELINE configs[2] and configs[3] exactly as bench.py times them: the 1M-triangle soup at 1920x1080 / 64 spp, and Cornell +
    glass + rough metal at 1920x1080 / 256 spp / max depth 16, 8 sample streams, default layouts, probed extend kernels — frames
    bit-identical to the oracle at the same spp and seed, ray counts equal (about a minute of oracle time on the box's host cores)."""
    sd = P.make_scene(P.native.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 0x5EED0001, 1920, 1080)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(1920, 1080, spp=64, max_depth=8, streams=8), 0)
    assert renderer.BvhInfo().width == 68
    assert_parity(img, st, ref, ost)
    assert st.paths == 1920 * 1080 * 64
    st2 = renderer.Render(0.0)  # the kernel the probe picked (lane-packing) for the whole frame
    assert int(st2.reserved[0]) == 2 and st2.rays == ost.rays and np.array_equal(renderer.ReadFramebuffer(), ref)
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 0x5EED0001, 1920, 1080)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(1920, 1080, spp=256, max_depth=16, streams=8), 0)
    assert renderer.BvhInfo().width == 2
    assert_parity(img, st, ref, ost)
    assert st.paths == 1920 * 1080 * 256


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline() == b"PF\n"
        w, h = (int(v) for v in f.readline().split())
        assert float(f.readline()) < 0  # little endian
        return np.frombuffer(f.read(), "<f4").reshape(h, w, 3)[::-1]  # rows are stored bottom-up


def test_native_host_cli(P, pto, renderer, tmp_path):
    """The compiled host over the C ABI (host/cpp/ptrt_cli: the stand-in for Program.cs:1-9 / App.cs:15-50 that this image can
    build) run as a program: one device, and three ranks through pt_comm on one device (--gpus 3 --virtual 1). Its PFM output must
    be the oracle's frame byte for byte, its PPM the UNORM8 image; the reference scene must be the reference's image."""
    import subprocess
    cli = os.path.join(os.path.dirname(HERE), "host", "cpp", "ptrt_cli")
    assert os.path.exists(cli), "host/cpp/ptrt_cli is built by __graft_entry__.build()"
    w, h, spp = 200, 131, 3
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 0x5EED0001, w, h)
    _, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(w, h, spp=spp, max_depth=8, streams=8), 0)  # the CLI's defaults
    want8 = renderer.ReadFramebufferRGBA8()
    for extra in ([], ["--gpus", "3", "--virtual", "1"], ["--gpus", "2", "--virtual", "2"]):
        pfm, ppm = str(tmp_path / "f.pfm"), str(tmp_path / "f.ppm")
        out = subprocess.run([cli, "--scene", "glass", "--size", f"{w}x{h}", "--spp", str(spp), "--pfm", pfm, "--ppm", ppm] + extra,
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.startswith(f"{ost.rays} rays"), (out.stdout, ost.rays, extra)
        got = _read_pfm(pfm)
        assert got.shape == (h, w, 3) and np.array_equal(got.view(np.uint32), np.ascontiguousarray(ref[..., :3]).view(np.uint32)), extra
        with open(ppm, "rb") as f:
            assert f.readline() == b"P6\n" and f.readline().split() == [str(w).encode(), str(h).encode()] and f.readline() == b"255\n"
            assert np.array_equal(np.frombuffer(f.read(), np.uint8).reshape(h, w, 3), want8[..., :3])
    ppm = str(tmp_path / "ref.ppm")
    out = subprocess.run([cli, "--scene", "reference", "--size", "320x200", "--ppm", ppm], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    _, b = pto.reference_sphere(320, 200)
    with open(ppm, "rb") as f:
        for _ in range(3):
            f.readline()
        assert np.array_equal(np.frombuffer(f.read(), np.uint8).reshape(200, 320, 3), b[..., :3])
    bad = subprocess.run([cli, "--scene", "glass", "--size", "0x0"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "ptrt status" in bad.stderr  # every failure is an exception, as in the reference (Renderer.cs:1022-1025)


def test_largest_slot_space(P, pto, renderer):
    """The largest frame the kernels' 32-bit slot offsets allow (2^28 slots = pixels x streams): 3840x2160 with 32 sample streams is
    265 M slots (a 4.2 GB float4 array: byte offsets just below 2^32). One sample per stream, depth 4, against the oracle; one stream
    more is refused."""
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_TESS, 200000, 0x5EED0001, 3840, 2160)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(3840, 2160, spp=32, max_depth=4, streams=32), 0)
    assert_parity(img, st, ref, ost)
    renderer.Params = P.make_params(3840, 2160, spp=33, max_depth=4, streams=33)
    with pytest.raises(P.PtException, match="too large"):
        renderer.Render(0.0)
    renderer.Params = P.make_params(64, 64, spp=1)  # a small frame gives the 20 GB of path state back (buffers shrink when 4x too big)
    renderer.Render(0.0)


def test_c4_at_its_real_depth_and_spp(P, pto, renderer):
    """BASELINE configs[3] (Cornell + glass + rough metal) at its real 256 spp and max depth 16, on a 480x270 frame."""
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 0x5EED0001, 480, 270)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(480, 270, spp=256, max_depth=16, streams=8), 0)
    assert renderer.BvhInfo().width == 2  # the default layout of a 12-triangle scene
    assert_parity(img, st, ref, ost)


def test_4k_frame_and_full_hd_soup(P, pto, renderer):
    """BASELINE configs[4] frame size (3840x2160, 1M-triangle Cornell) and configs[2] (1M-triangle soup, 1080p) at 1 spp,
    8 streams allocated: the largest slot spaces the benchmark configurations use, against the oracle."""
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_TESS, 1 << 20, 0x5EED0001, 3840, 2160)
    p4k = P.make_params(3840, 2160, spp=1, max_depth=8, streams=8)
    img, st, ref, ost = run_both(P, pto, renderer, sd, p4k, 0)
    assert_parity(img, st, ref, ost)
    assert st.paths == 3840 * 2160 and (img[..., 3] == 1.0).all()
    # configs[4]'s partition exactly — the 4K frame's 2,040 tiles over EIGHT ranks, gathered and un-tiled through pt_comm — as far as one
    # GPU can take it (virtual ranks): the assembled frame is the oracle's frame, every rank traced its share
    with P.Comm([renderer] * 8) as comm:
        stats = comm.Render(p4k)
        assert sum(s.rays for s in stats) == ost.rays and min(s.rays for s in stats) > 0.1 * ost.rays
        assert np.array_equal(renderer.ReadFramebuffer(), ref)
    sd = P.make_scene(P.native.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 0x5EED0001, 1920, 1080)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(1920, 1080, spp=1, max_depth=8, streams=8), 0)
    assert_parity(img, st, ref, ost)


@pytest.mark.parametrize("layout", [2, 4, 68, 72, 73])
def test_gpu_lbvh_builder(P, pto, renderer, layout):
    """hierarchy built on the GPU (Morton sort + Karras + refit). The blob must pass the oracle's structural
    validator, the HIP frame must equal the oracle traversing THOSE bytes (rays and visit counts included), and — because the
    closest hit does not depend on the tree (SPEC §4) — it must equal the frame of the host-SAH-built scene bit for bit."""
    N = P.native
    for kind, detail in ((N.PT_SCENE_TRIANGLE_SOUP, 40000), (N.PT_SCENE_CORNELL_TESS, 30000), (N.PT_SCENE_CORNELL, 0)):
        sd = P.make_scene(kind, detail, 11, 160, 100)
        p = P.make_params(160, 100, spp=3, max_depth=6, streams=2)
        sah_img, sah_st, _, _ = run_both(P, pto, renderer, sd, p, layout)
        img, st, ref, ost = run_both(P, pto, renderer, sd, p, layout | N.PT_BVH_BUILD_LBVH, count=True)  # run_both validates the blob
        assert_parity(img, st, ref, ost)
        assert (st.node_visits, st.tri_tests) == (ost.node_visits, ost.tri_tests)
        assert st.rays == sah_st.rays and np.array_equal(img, sah_img)
    # degenerate input: every centroid identical (all Morton codes equal) still yields a valid, bounded-depth tree
    sd = P.make_scene(N.PT_SCENE_TRIANGLE_SOUP, 3000, 11, 64, 64)
    sd.verts[:] = sd.verts[0]
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(64, 64, spp=1, max_depth=3), layout | N.PT_BVH_BUILD_LBVH)
    assert_parity(img, st, ref, ost)
    assert renderer.BvhInfo().max_depth <= 16


def test_rank_partition_is_image_invariant(P, pto, renderer):
    """SPEC §6: the picture must not depend on the number of ranks. R 'virtual ranks' rendered one after the other on this GPU and
    assembled through the library's own exchange entry points (pt_comm_*) == the single-rank frame, bit for bit — once with
    pt_comm_render, once with the caller driving pt_render + pt_comm_stage_tiles + pt_comm_assemble, and once by hand through
    pt_tiles_device_ptr / pt_assemble_tiles (what a process-per-GPU host does around its own collective)."""
    import torch
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 1, 200, 131)
    renderer.SetScene(sd, 4)
    renderer.Params = P.make_params(200, 131, spp=3, max_depth=6)
    one = renderer.Render(0.0)
    want = renderer.ReadFramebuffer()
    for nranks in (2, 3, 5):
        with P.Comm([renderer] * nranks, root=nranks - 1) as comm:
            stats = comm.Render(P.make_params(200, 131, spp=3, max_depth=6))
            assert sum(s.rays for s in stats) == one.rays and sum(s.paths for s in stats) == one.paths
            assert np.array_equal(renderer.ReadFramebuffer(), want), nranks
            for rank in range(nranks):  # the same, driven from outside
                renderer.Params = P.make_params(200, 131, spp=3, max_depth=6, rank=rank, nranks=nranks)
                renderer.Render(0.0)
                comm.StageTiles(rank)
            comm.Assemble(P.make_params(200, 131, spp=3, max_depth=6))
            assert np.array_equal(renderer.ReadFramebuffer(), want), nranks
            with pytest.raises(P.PtException, match="not staged"):
                comm.Assemble(P.make_params(200, 131, spp=3, max_depth=6))
        blocks, rays = [], 0
        for rank in range(nranks):
            renderer.Params = P.make_params(200, 131, spp=3, max_depth=6, rank=rank, nranks=nranks)
            st = renderer.Render(0.0)
            rays += st.rays
            blocks.append(torch.as_tensor(renderer.TilesDevice(), device="cuda").clone())
        gathered = torch.cat(blocks)
        torch.cuda.synchronize()
        renderer.AssembleTiles(gathered.data_ptr(), gathered.numel())
        assert rays == one.rays
        assert np.array_equal(renderer.ReadFramebuffer(), want), nranks


def test_comm_rccl_path_on_one_gpu(P, pto, renderer):
    """pt_comm with PT_COMM_FORCE_RCCL and one rank: librccl is dlopen-ed, ncclCommInitAll makes a one-rank communicator and the
    frame goes through ncclGather + un-tiling — the code path of the multi-GPU exchange, as far as one GPU can take it."""
    N = P.native
    sd = P.make_scene(N.PT_SCENE_CORNELL_TESS, 20000, 3, 300, 170)
    renderer.SetScene(sd, 0)
    p = P.make_params(300, 170, spp=4, max_depth=6, streams=2)
    renderer.Params = p
    one = renderer.Render(0.0)
    want = renderer.ReadFramebuffer()
    with P.Comm([renderer], flags=N.PT_COMM_FORCE_RCCL) as comm:
        for _ in range(2):
            stats = comm.Render(p)
            assert stats[0].rays == one.rays and np.array_equal(renderer.ReadFramebuffer(), want)
    with pytest.raises(P.PtException):
        P.Comm([renderer], root=1)


def test_comm_with_a_context_per_rank_on_one_gpu(P, pto, renderer):
    """pt_comm's multi-context branch as far as one GPU can take it (PT_COMM_COPY_EXCHANGE): three ranks, each with its OWN context
    (own streams, queues, tile buffer) on device 0, rendered concurrently by one host thread per context inside pt_comm_render; the
    tile blocks reach the root by device copies on the ranks' own streams; un-tiled on the root. Frame = the single-context frame bit
    for bit, also with the root in the middle, also when the comm outlives nothing and is destroyed after its contexts."""
    N = P.native
    w, h = 333, 190
    sd = P.make_scene(N.PT_SCENE_CORNELL_TESS, 20000, 3, w, h)
    p = P.make_params(w, h, spp=5, max_depth=6, streams=2)
    renderer.SetScene(sd, 0)
    renderer.Params = p
    one = renderer.Render(0.0)
    want = renderer.ReadFramebuffer()
    rs = [P.Renderer(P.Window(w, h)) for _ in range(3)]
    try:
        for r in rs:
            r.Init()
            r.SetScene(sd, 0)
        with pytest.raises(P.PtException):  # without the flag distinct contexts on one device are refused (they would need RCCL ranks on one GPU)
            P.Comm(rs)
        for root in (0, 1):
            with P.Comm(rs, root=root, flags=N.PT_COMM_COPY_EXCHANGE) as comm:
                for _ in range(2):
                    stats = comm.Render(p)
                    assert sum(s.rays for s in stats) == one.rays
                    assert np.array_equal(rs[root].ReadFramebuffer(), want), root
        comm = P.Comm(rs, flags=N.PT_COMM_COPY_EXCHANGE)
        comm.Render(p)
    finally:
        for r in rs:
            r.Dispose()
    comm.Dispose()  # after its contexts: pt_comm_destroy does not touch them


def test_rccl_gather_aliases_the_tile_buffer(P, pto, renderer):
    """What bench.py does between frames for N > 1, as far as one GPU can take it: a torch.distributed NCCL (= RCCL) process group
    (one rank), the library's tile buffer aliased through __cuda_array_interface__ as the gather's send buffer (no copy), the stream
    synchronised before the next frame, the gathered buffer un-tiled by pt_assemble_tiles."""
    import os
    import torch
    import torch.distributed as dist
    from pathtracing_amd.distributed import gather_tiles
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_TESS, 30000, 5, 321, 200)
    renderer.SetScene(sd, 0)
    p = P.make_params(321, 200, spp=3, max_depth=6, streams=2)
    renderer.Params = p
    renderer.Render(0.0)
    want = renderer.ReadFramebuffer()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29671")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        lay = P.tile_layout(p)
        per_rank = lay.tiles_per_rank * lay.floats_per_tile
        recv = torch.zeros(per_rank, dtype=torch.float32, device="cuda")
        for _ in range(2):
            renderer.Render(0.0)
            mine = torch.as_tensor(renderer.TilesDevice(), device="cuda")
            assert mine.numel() == per_rank
            dist.gather(mine, [recv], dst=0)  # gather_tiles() short-cuts world == 1; this is the collective it issues otherwise
            torch.cuda.current_stream().synchronize()
            renderer.AssembleTiles(recv.data_ptr(), recv.numel())
            assert np.array_equal(renderer.ReadFramebuffer(), want)
        assert gather_tiles(mine, per_rank, 0, 1, dist) is mine
    finally:
        dist.destroy_process_group()


def test_bench_exchange_path_on_one_gpu():
    """bench.py --force-exchange: the N > 1 frame loop (RCCL process group, PipelinedGather: staging copy, gather on a side stream
    overlapped with the next frame, deferred un-tiling, double buffering) with a group of one rank; bench.py itself asserts that the
    assembled frame equals the plain frame. Also run under a 2-rank gloo group (CPU-staged gather) as the driver launches N > 1."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--scene", "cornell_tess", "--tris", "40000", "--width", "640", "--height", "360", "--spp", "8", "--steps", "5", "--warmup", "2",
              "--no-cpu-baseline", "--no-roofline", "--no-configs"]
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-exchange"] + common, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["parallelism"] == "tiles1-exchange-forced" and line["value"] > 0
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29731",
                          os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-gloo"] + common, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    rk = line["ranks"]  # what every rank did: wall time, rays, kernel time per frame; rays add up to the frame
    assert len(rk["wall_s"]) == 2 and sum(rk["rays_per_frame"]) == line["config"]["rays_per_frame"] and rk["gpu_ms_max_over_mean"] >= 1.0
    # four ranks (with the test's own context: five processes on the card, the most a box allows short of its limit of six): the frame
    # split four ways, gathered and assembled as the driver's N = 4 run would; bench.py asserts the assembled frame is the one-rank frame
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", "29733",
                          os.path.join(root, "bench.py"), "--gpus", "4", "--rehearse-gloo"] + common, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 4 and len(line["ranks"]["gpu_ms_per_frame"]) == 4 and sum(line["ranks"]["rays_per_frame"]) == line["config"]["rays_per_frame"]


@pytest.mark.parametrize("streams", [2, 4, 7, 40])
def test_sample_streams(P, pto, renderer, streams):
    """SPEC §5: K sample streams per pixel in flight, each with its own partial sum, summed in fixed order.
    The oracle forms the same partial sums, so the picture stays bit-identical for every K (also K > spp, K not dividing spp)."""
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 5, 150, 100)
    for spp in (1, 5, 8):
        img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(150, 100, spp=spp, max_depth=10, streams=streams), 68)
        assert_parity(img, st, ref, ost)
    # and the tile partition still does not change the picture
    import torch
    renderer.Params = P.make_params(150, 100, spp=6, max_depth=6, streams=streams)
    one = renderer.Render(0.0); want = renderer.ReadFramebuffer()
    blocks = []
    for rank in range(2):
        renderer.Params = P.make_params(150, 100, spp=6, max_depth=6, streams=streams, rank=rank, nranks=2)
        renderer.Render(0.0)
        blocks.append(torch.as_tensor(renderer.TilesDevice(), device="cuda").clone())
    g = torch.cat(blocks); torch.cuda.synchronize()
    renderer.AssembleTiles(g.data_ptr(), g.numel())
    assert np.array_equal(renderer.ReadFramebuffer(), want)


def test_progressive_accumulation(P, pto, renderer):
    """PT_FLAG_ACCUMULATE (the analogue of the reference's per-frame loop App.cs:39-42): three calls of
    3 + 5 + 2 samples show exactly the frame of one 10-sample call, for 1 and 4 streams."""
    N = P.native
    sd = P.make_scene(N.PT_SCENE_CORNELL_GLASS, 0, 9, 120, 90)
    renderer.SetScene(sd, 0)
    for streams in (1, 4):
        done = 0
        for n in (3, 5, 2):
            renderer.Params = P.make_params(120, 90, spp=n, max_depth=8, streams=streams, sample_offset=done,
                                            flags=N.PT_FLAG_ACCUMULATE if done else 0)
            renderer.Render(0.0)
            done += n
        img = renderer.ReadFramebuffer()
        info = renderer.BvhInfo()
        ref, _ = pto.render(pto.Scene(sd, (info.width,) + renderer.BvhRead()), P.make_params(120, 90, spp=10, max_depth=8, streams=streams))
        assert np.array_equal(img, ref)
    # a frame that does not continue the previous one is refused
    renderer.Params = P.make_params(120, 90, spp=2, streams=4, sample_offset=3, flags=N.PT_FLAG_ACCUMULATE)
    with pytest.raises(P.PtException, match="sample_offset"):
        renderer.Render(0.0)
    renderer.Params = P.make_params(64, 64, spp=2, streams=4, sample_offset=10, flags=N.PT_FLAG_ACCUMULATE)
    with pytest.raises(P.PtException, match="previous frame"):
        renderer.Render(0.0)


def test_edge_cases(P, pto, renderer):
    N = P.native
    cam = P.make_scene(0, 0, 0, 70, 40).cam
    # empty scene: every path is one ray into the sky
    sd = P.SceneData(cam=cam); sd.sky = np.array([0.25, 0.5, 1.0], np.float32)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(70, 40, spp=2, max_depth=4))
    assert_parity(img, st, ref, ost) and None
    assert st.rays == 70 * 40 * 2 and np.allclose(img[..., :3], [0.25, 0.5, 1.0])
    # one triangle, no spheres; max_depth 1 (camera rays only); spp 1
    sd = P.SceneData(cam=cam)
    sd.verts = np.array([[-1, -1, 0, 1, -1, 0, 0, 1, 0]], np.float32); sd.tri_mat = np.zeros(1, np.uint32)
    m = np.zeros(1, P.MATERIAL_DTYPE); m["albedo"] = 0.5; m["emission"] = (1, 2, 3); sd.mats = m
    for width in (2, 4, 68, 72, 73):
        img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(70, 40, spp=1, max_depth=1), width)
        assert_parity(img, st, ref, ost)
        assert st.rays == 70 * 40 and img[..., 0].max() == 1.0
    # spheres only (no BVH nodes at all)
    sd = P.make_scene(N.PT_SCENE_CORNELL_GLASS, 0, 0, 70, 40); sd.verts = sd.verts[:0]; sd.tri_mat = sd.tri_mat[:0]; sd.sky[:] = 1
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(70, 40, spp=4, max_depth=12))
    assert_parity(img, st, ref, ost)
    # sample_offset continues the RNG stream: frame(offset=2, spp=2) equals samples 2,3 of an spp=4 frame
    sd = P.make_scene(N.PT_SCENE_CORNELL, 0, 0, 70, 40)
    img, st, ref, ost = run_both(P, pto, renderer, sd, P.make_params(70, 40, spp=2, max_depth=5, sample_offset=2))
    assert_parity(img, st, ref, ost)


def test_errors_raise_like_the_reference(P, renderer):
    """Every failure is an exception (the reference throws on every Vulkan failure, e.g. Renderer.cs:1022-1025)."""
    N = P.native
    sd = P.make_scene(0, 0, 0, 64, 64)
    renderer.SetScene(sd, 0)
    for kw in (dict(spp=0), dict(max_depth=0), dict(max_depth=300), dict(ray_eps=float("nan")), dict(mode=7)):
        renderer.Params = P.make_params(64, 64, **kw)
        with pytest.raises(P.PtException):
            renderer.Render(0.0)
    renderer.Params = P.make_params(64, 64, nranks=2, rank=0)
    renderer.Render(0.0)
    with pytest.raises(P.PtException, match="no assembled frame"):
        renderer.ReadFramebuffer()
    r2 = P.Renderer(P.Window(64, 64)); r2.Init(); r2.Params = P.make_params(64, 64)
    with pytest.raises(P.PtException):          # no scene set
        r2.Render(0.0)
    r2.Dispose(); r2.Dispose()                  # idempotent, like Dispose(bool) at Renderer.cs:1192


def test_bucketed_specular_shading_is_identical(P, pto, renderer):
    """PT_FLAG_BUCKET_SPECULAR: metal / dielectric hits shaded from per-kind bucket queues instead of in queue order.
    A scheduling choice only — frame, ray count and visit counters must equal the oracle's either way."""
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 21, 180, 120)
    for flags in (0, P.native.PT_FLAG_BUCKET_SPECULAR):
        p = P.make_params(180, 120, spp=6, max_depth=16, streams=2, flags=flags)
        img, st, ref, ost = run_both(P, pto, renderer, sd, p, 0, count=True)
        assert_parity(img, st, ref, ost)
        assert (st.node_visits, st.tri_tests, st.sphere_tests) == (ost.node_visits, ost.tri_tests, ost.sphere_tests)


def test_shard_groups_on_separate_streams(P, pto, monkeypatch):
    """pt_tuning.loops = 2/4: the 64 queue shards run as 2/4 independent wavefront loops on their own HIP streams (api.cpp).
    Shards never exchange slots, so the frame and the ray count must not change."""
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 4, 300, 200)
    p = P.make_params(300, 200, spp=6, max_depth=10, streams=4)
    frames = []
    for groups in ("1", "2", "4"):
        r = P.Renderer(P.Window(300, 200)); r.Init()
        r.SetTuning(loops=int(groups))
        try:
            r.SetScene(sd, 0); r.Params = p
            st = r.Render(0.0)
            frames.append((st.rays, r.ReadFramebuffer()))
            if groups == "2":
                info = r.BvhInfo()
                ref, ost = pto.render(pto.Scene(sd, (info.width,) + r.BvhRead()), p)
                assert ost.rays == st.rays and np.array_equal(frames[-1][1], ref)
        finally:
            r.Dispose()
    assert all(f[0] == frames[0][0] and np.array_equal(f[1], frames[0][1]) for f in frames)


@pytest.mark.parametrize("flags", [0, 4, 128, 64, 68])  # fused one-ray-per-lane / lane-packing / pooled kernels, and the first two followed by k_shade (PT_FLAG_SPLIT_KERNELS)
def test_queue_compaction_policy_is_invisible(P, pto, monkeypatch, flags):
    """pt_tuning.compact_below / sticky_samples: queues carried over in place with holes (0 = never re-packed), re-packed every
    launch (2), re-packed when the predicted alive/length ratio is < 0.9 or 0.5, and the short-frame rule (2 samples per stream:
    every launch) — scheduling only: frame and ray count stay the oracle's. The streams end at different iterations (spp 7 over
    4 streams, Russian roulette), so the in-place runs do see holes."""
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 4, 200, 150)
    p = P.make_params(200, 150, spp=7, max_depth=12, streams=4, flags=flags)
    frames = []
    for thr, sticky in ((0.0, 0), (2.0, 0), (0.9, 0), (0.5, 0), (0.9, 32), (0.0, 32)):
        r = P.Renderer(P.Window(200, 150)); r.Init()
        r.SetTuning(compact_below=thr, sticky_samples=sticky)
        assert r.GetTuning().sticky_samples == sticky
        try:
            r.SetScene(sd, 0); r.Params = p
            st = r.Render(0.0)
            frames.append((st.rays, r.ReadFramebuffer(), st.reserved[1], st.iterations))
            if thr == 0.0 and sticky == 0:
                info = r.BvhInfo()
                ref, ost = pto.render(pto.Scene(sd, (info.width,) + r.BvhRead()), p)
                assert ost.rays == st.rays and np.array_equal(frames[-1][1], ref)
        finally:
            r.Dispose()
    assert all(f[0] == frames[0][0] and np.array_equal(f[1], frames[0][1]) for f in frames)
    assert frames[0][2] == 0 and frames[1][2] >= frames[2][2] > 0  # (shard, iteration) pairs that re-packed (the lane-packing kernel ends this small frame in one)
    assert frames[4][2] == frames[1][2] and frames[5][2] == 0      # short frame: every launch; compact_below 0 switches that off too


@pytest.mark.parametrize("flags", [8, 128])
def test_sticky_repacking_of_short_frames(P, pto, flags):
    """A frame of few samples per stream (here 3) re-packs a shard's queue in every launch once the shard has re-packed at all;
    with sticky_samples = 0 only the predicted ratio decides. Same picture, more re-packs."""
    sd = P.make_scene(P.native.PT_SCENE_CORNELL, 0, 9, 256, 160)
    p = P.make_params(256, 160, spp=12, max_depth=8, streams=4, flags=flags)
    out = []
    for sticky in (0, 32):
        r = P.Renderer(P.Window(256, 160)); r.Init()
        r.SetTuning(sticky_samples=sticky, finish_below=0, bounces=2)
        try:
            r.SetScene(sd, 0); r.Params = p
            st = r.Render(0.0)
            out.append((st.rays, r.ReadFramebuffer(), st.reserved[1]))
            if sticky == 0:
                info = r.BvhInfo()
                ref, ost = pto.render(pto.Scene(sd, (info.width,) + r.BvhRead()), p)
                assert ost.rays == st.rays and np.array_equal(out[-1][1], ref)
        finally:
            r.Dispose()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    assert out[1][2] > out[0][2] > 0


@pytest.mark.parametrize("kflag", [8, 128])  # one ray per lane, pooled
def test_bounces_per_launch_is_invisible(P, pto, monkeypatch, kflag):
    """pt_tuning.bounces: the fused kernel advances a path by 1, 3 or 16 vertices per launch with its state in registers
    (kernels.hip k_extend). Same arithmetic per vertex, so frame, ray count and visit counters stay the oracle's."""
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 4, 160, 120)
    p = P.make_params(160, 120, spp=5, max_depth=9, streams=2, flags=P.native.PT_FLAG_COUNT_VISITS | kflag)
    got = []
    # pt_tuning.finish_below: shards with no more alive paths than this run them to their end in one launch (0 = never, 1000000 = always)
    for b, finish in (("1", "0"), ("3", "0"), ("16", "0"), ("2", "1000000")):
        r = P.Renderer(P.Window(160, 120)); r.Init()
        r.SetTuning(bounces=int(b), finish_below=int(finish))
        assert r.GetTuning().bounces == int(b)
        with pytest.raises(P.PtException):
            r.SetTuning(loops=3)
        try:
            r.SetScene(sd, 0); r.Params = p
            st = r.Render(0.0)
            got.append((st.rays, st.node_visits, st.tri_tests, st.sphere_tests, r.ReadFramebuffer(), st.iterations))
            if b == "3":
                info = r.BvhInfo()
                ref, ost = pto.render(pto.Scene(sd, (info.width,) + r.BvhRead()), p)
                assert (ost.rays, ost.node_visits, ost.tri_tests, ost.sphere_tests) == got[-1][:4] and np.array_equal(got[-1][4], ref)
        finally:
            r.Dispose()
    assert all(g[:4] == got[0][:4] and np.array_equal(g[4], got[0][4]) for g in got)
    assert got[0][5] > got[1][5] > got[2][5]  # fewer launches


def test_image_output(P, pto, renderer, tmp_path):
    """PPM carries the reference's R8G8B8A8Unorm quantisation (Renderer.cs:124), PFM the linear floats."""
    renderer.Params = P.make_params(97, 41, mode=P.native.PT_REFERENCE_SPHERE)
    renderer.Render(0.0)
    renderer.SaveImage(str(tmp_path / "f.ppm")); renderer.SaveImage(str(tmp_path / "f.pfm"))
    of, ob = pto.reference_sphere(97, 41)
    raw = open(tmp_path / "f.ppm", "rb").read()
    assert raw.startswith(b"P6\n97 41\n255\n") and np.array_equal(np.frombuffer(raw[len(b"P6\n97 41\n255\n"):], np.uint8).reshape(41, 97, 3), ob[..., :3])
    # what the reference's window shows: the UNORM8 image through its sRGB swapchain (SwapChain.cs:157-158), alpha linear
    s8 = renderer.ReadFramebufferSRGB8()
    lin = ob.astype(np.float64) / 255.0
    enc = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * lin ** (1 / 2.4) - 0.055)
    want = np.floor(255.0 * enc + 0.5).astype(np.uint8)
    assert np.array_equal(s8[..., :3], want[..., :3]) and np.array_equal(s8[..., 3], ob[..., 3])
    renderer.SaveImage(str(tmp_path / "s.ppm"), srgb=True)
    raw = open(tmp_path / "s.ppm", "rb").read()
    assert np.array_equal(np.frombuffer(raw[len(b"P6\n97 41\n255\n"):], np.uint8).reshape(41, 97, 3), want[..., :3])
    raw = open(tmp_path / "f.pfm", "rb").read()
    hdr = b"PF\n97 41\n-1.0\n"
    assert raw.startswith(hdr) and np.array_equal(np.frombuffer(raw[len(hdr):], "<f4").reshape(41, 97, 3)[::-1], of[..., :3])


def test_app_runs_the_reference_frame(P, pto):
    """Program.cs:3-7 / App.Run (App.cs:15-21): window 1920x1080, renderer, render loop — one frame of the reference kernel."""
    with P.App(frames=2) as app:
        app.Run()
        b = app.Renderer.ReadFramebufferRGBA8()
    assert b.shape == (1080, 1920, 4) and int((b[..., 2] > 0).sum()) == 305317


def test_host_readback_and_kernel_pin(P, pto):
    """pt_tuning.readback / lag / extend_kernel (ABI 2). How a launch's queue sizes reach the host — stored to mapped pinned memory by
    the next launch's first threads (default) or copied behind every launch — and how far the host runs ahead are scheduling only:
    same frame, same rays, for every combination, on long frames and on frames that are over within the run-ahead. A pinned extend
    kernel is the one that runs (pt_stats.reserved[0]) unless a frame's flag overrides it; out-of-range values are refused."""
    N = P.native
    sd = P.make_scene(N.PT_SCENE_CORNELL_TESS, 6000, 9, 260, 150)
    r = P.Renderer(P.Window(260, 150)); r.Init()
    try:
        r.SetScene(sd, 0)
        info = r.BvhInfo()
        osc = pto.Scene(sd, (info.width,) + r.BvhRead())
        for spp, depth in ((1, 2), (9, 8), (40, 6)):
            p = P.make_params(260, 150, spp=spp, max_depth=depth, streams=4)
            ref, ost = pto.render(osc, p)
            for readback in (0, 1):
                for lag in (0, 2, 3, 5):
                    for loops in (1, 2):
                        r.SetTuning(readback=readback, lag=lag, loops=loops, extend_kernel=1)
                        r.Params = p
                        for _ in range(2):  # twice: the second frame starts from the cached queue / counter template
                            st = r.Render(0.0)
                            assert st.rays == ost.rays and np.array_equal(r.ReadFramebuffer(), ref), (spp, readback, lag, loops)
        p = P.make_params(260, 150, spp=9, max_depth=8, streams=4)
        for pin in (1, 2, 3):
            r.SetTuning(readback=0, lag=0, loops=0, extend_kernel=pin)
            r.Params = p
            assert int(r.Render(0.0).reserved[0]) == pin
        r.Params = P.make_params(260, 150, spp=9, max_depth=8, streams=4, flags=N.PT_FLAG_EXTEND_SIMPLE)
        assert int(r.Render(0.0).reserved[0]) == 1  # a frame's flag beats the context's pin
        for bad in (dict(extend_kernel=4), dict(readback=2), dict(lag=1), dict(lag=6)):
            with pytest.raises(P.PtException):
                r.SetTuning(**bad)
        # frames too small to time settle on the one-ray-per-lane kernel instead of probing for ever (one loop, no finish mode)
        r.SetTuning(extend_kernel=0)
        r.SetScene(sd, 0)
        r.Params = P.make_params(64, 64, spp=2, max_depth=4, streams=2)
        seen = [int(r.Render(0.0).reserved[0]) for _ in range(6)]
        assert seen[-1] == 1 and seen[-2] == 1
    finally:
        r.Dispose()


def test_frame_start_template_follows_the_geometry(P, pto, renderer):
    """The first queue and counter block of a frame are cached per frame geometry (api.cpp: q_init / cnt_init). Changing size, rank
    split, stream count, a sample count below the stream count, or the sample offset must rebuild them: every frame of the
    sequence — including returns to an earlier geometry and progressive frames that continue one — equals the oracle's."""
    N = P.native
    sd = P.make_scene(N.PT_SCENE_CORNELL_GLASS, 0, 12, 200, 120)
    renderer.SetScene(sd, 0)
    info = renderer.BvhInfo()
    osc = pto.Scene(sd, (info.width,) + renderer.BvhRead())
    seq = [(200, 120, 6, 4, 0), (200, 120, 6, 4, 0), (130, 97, 6, 4, 0), (130, 97, 2, 4, 0), (130, 97, 2, 4, 3), (130, 97, 6, 8, 0),
           (200, 120, 6, 4, 0), (64, 64, 1, 1, 0), (200, 120, 3, 4, 1)]
    for w, h, spp, streams, offset in seq:
        sdw = P.make_scene(N.PT_SCENE_CORNELL_GLASS, 0, 12, w, h)  # the camera follows the aspect ratio
        renderer.SetScene(sdw, 0)
        p = P.make_params(w, h, spp=spp, max_depth=7, streams=streams, sample_offset=offset)
        renderer.Params = p
        st = renderer.Render(0.0)
        ref, ost = pto.render(pto.Scene(sdw, (renderer.BvhInfo().width,) + renderer.BvhRead()), p)
        assert st.rays == ost.rays and np.array_equal(renderer.ReadFramebuffer(), ref), (w, h, spp, streams, offset)
    # progressive: 3 + 2 + 4 samples continue one set of partial sums across the cached frame start (the stream phase changes each time)
    sdw = P.make_scene(N.PT_SCENE_CORNELL_GLASS, 0, 12, 200, 120)
    renderer.SetScene(sdw, 0)
    whole = P.make_params(200, 120, spp=9, max_depth=7, streams=4)
    ref, _ = pto.render(pto.Scene(sdw, (renderer.BvhInfo().width,) + renderer.BvhRead()), whole)
    done = 0
    for k, n in enumerate((3, 2, 4)):
        renderer.Params = P.make_params(200, 120, spp=n, max_depth=7, streams=4, sample_offset=done, flags=N.PT_FLAG_ACCUMULATE if k else 0)
        renderer.Render(0.0)
        done += n
    assert np.array_equal(renderer.ReadFramebuffer(), ref)
    del osc


@pytest.mark.parametrize("kind,depth,spp,n,own_roulette,seed", [(0, 8, 16384, 30000, False, 77), (1, 16, 32768, 40000, True, 4242)])
def test_hip_frames_match_the_independent_estimator(P, renderer, kind, depth, spp, n, own_roulette, seed):
    """The physics pin of tests/test_physics_pin.py applied to the HIP path itself, without the oracle in between: converged 4 x 4
    frames of the Cornell box and of C4's glass + rough-metal scene FROM THE DEVICE against the float64 numpy tracer that shares no
    sampling routine, RNG, weight formula or intersection code with it — every channel within the same bands (3 sigma, chi-square)."""
    from test_physics_pin import numpy_radiance
    w = h = 4
    sd = P.make_scene(kind, 0, 0x5EED0001, w, h)
    renderer.SetScene(sd, 0)
    renderer.Params = P.make_params(w, h, spp=spp, max_depth=depth, rr_start=3, seed=seed, streams=8)
    renderer.Render(0.0)
    img = renderer.ReadFramebuffer()
    mean, err = numpy_radiance(sd, w, h, n, depth, 3, np.random.default_rng(2026 + kind), own_roulette=own_roulette)
    sigma = np.sqrt(err ** 2 + (err * np.sqrt(n / spp)) ** 2)
    z = np.abs(img[..., :3] - mean) / np.maximum(sigma, 1e-6)
    assert (z > 3.0).sum() <= 3 and z.max() < 5.0, (z.max(), (z > 3).sum())
    assert (z * z).sum() < 48 + 6 * np.sqrt(2 * 48), (z * z).sum()
    assert img[..., :3].mean() > 0.05 and (img[..., 3] == 1.0).all()


def test_randomised_configurations_against_the_oracle(P, pto, renderer):
    """A fixed-seed sweep over scene kind, frame size (ragged tiles), spp, depth, streams, node layout, extend kernel, pipeline,
    sample offset and the scheduling knobs (pt_tuning: loops, bounces per launch, re-packing threshold and sticky limit, run-to-end
    threshold): every combination must reproduce the oracle's frame, ray count and visit counters bit for bit. Guards the
    interplay of the scheduling features (in-place queues, predicted / sticky / forced re-packing, run-to-end tail, shard loops)."""
    N = P.native
    rng = np.random.default_rng(20261005)
    kinds = [(N.PT_SCENE_CORNELL, 0), (N.PT_SCENE_CORNELL_GLASS, 0), (N.PT_SCENE_TRIANGLE_SOUP, 3000), (N.PT_SCENE_CORNELL_TESS, 2500)]
    defaults = renderer.GetTuning()
    try:
        for case in range(72):
            kind, detail = kinds[int(rng.integers(len(kinds)))]
            w, h = int(rng.integers(1, 150)), int(rng.integers(1, 110))
            spp, depth = int(rng.integers(1, 12)), int(rng.integers(1, 14))
            if case % 9 == 8:
                spp = int(rng.integers(34, 48))  # many samples per stream: the predicted-ratio rule, not the sticky one
            streams = int(rng.choice([0, 1, 2, 3, 8, 16]))
            width = int(rng.choice([0, 2, 4, 68, 72, 73]))
            flags = int(rng.choice([0, 0, N.PT_FLAG_EXTEND_PACKED, N.PT_FLAG_EXTEND_SIMPLE, N.PT_FLAG_EXTEND_POOL, N.PT_FLAG_EXTEND_POOL,
                                    N.PT_FLAG_SPLIT_KERNELS, N.PT_FLAG_SPLIT_KERNELS | N.PT_FLAG_EXTEND_PACKED, N.PT_FLAG_BUCKET_SPECULAR]))
            offset = int(rng.integers(0, 5))
            tune = dict(loops=int(rng.choice([0, 1, 2, 4])), bounces=int(rng.choice([0, 0, 1, 2, 3, 5, 8])),
                        compact_below=float(rng.choice([0.0, 0.5, 0.9, 0.9, 1.0, 2.0])), sticky_samples=int(rng.choice([0, 2, 32, 32, 1000])),
                        finish_below=int(rng.choice([0, 64, 4096, 4096, 1 << 20])), lag=int(rng.choice([0, 0, 2, 3, 4, 5])),
                        readback=int(rng.choice([0, 0, 1])), extend_kernel=int(rng.choice([0, 0, 0, 1, 2, 3])))
            renderer.SetTuning(**tune)
            sd = P.make_scene(kind, detail, int(rng.integers(1, 1 << 30)), w, h)
            p = P.make_params(w, h, spp=spp, max_depth=depth, streams=streams, flags=flags, sample_offset=offset, seed=int(rng.integers(1 << 31)))
            count = case % 2 == 0  # the counting and the plain instantiations of the kernels are different code objects: cover both
            img, st, ref, ost = run_both(P, pto, renderer, sd, p, width, count=count)
            ctx = (case, kind, w, h, spp, depth, streams, width, flags, offset, tune)
            assert st.rays == ost.rays and st.paths == ost.paths, ctx
            assert np.array_equal(img, ref), ctx
            if count:
                assert (st.node_visits, st.tri_tests, st.sphere_tests) == (ost.node_visits, ost.tri_tests, ost.sphere_tests), ctx
    finally:
        renderer.SetTuning(**{k: getattr(defaults, k) for k, _ in defaults._fields_})


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [8, 4])  # one ray per lane, lane-packing
def test_sphere_lists_of_any_length(P, pto, renderer, flags):
    """The kernels read the sphere list four spheres per scalar load (device arrays padded to a multiple of four): lists of
    0, 1, 2, 3, 5, 6 and 9 spheres must give the oracle's frame, ray count and sphere-test count."""
    base = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 3, 160, 120)
    rng = np.random.default_rng(5)
    pool = np.concatenate([base.spheres] + [base.spheres * np.float32([1, 1, 1, 0.6]) + np.float32([dx, 0.35 * (k + 1), dz, 0]) for k, (dx, dz) in
                                            enumerate(rng.uniform(-0.15, 0.15, (2, 2)))])[:9].astype(np.float32)
    pool_mat = np.concatenate([base.sph_mat] * 3)[:9].astype(np.uint32)
    for n in (0, 1, 2, 3, 5, 6, 9):
        sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 3, 160, 120)
        sd.spheres, sd.sph_mat = pool[:n].copy(), pool_mat[:n].copy()
        p = P.make_params(160, 120, spp=6, max_depth=10, streams=2, flags=flags)
        img, st, ref, ost = run_both(P, pto, renderer, sd, p, 0, count=True)
        assert st.rays == ost.rays and np.array_equal(img, ref), n
        assert (st.node_visits, st.tri_tests, st.sphere_tests) == (ost.node_visits, ost.tri_tests, ost.sphere_tests), n
        assert st.sphere_tests == n * st.rays
