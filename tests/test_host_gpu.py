"""GPU tests through the C++ host mirror (same names as src/passes/draw_gen.rs / cluster.rs):
the reference's frame sequence — early cull, depth pyramid, late cull, light clusters — against
the CPU oracle."""
import numpy as np
import pytest

import scenes as sc
from orbit_amd import layouts as L, passes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available()
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()


def fetch(buf, nbytes):
    return passes.buffer_tensor(buf)[:nbytes].cpu().numpy()


@pytest.mark.parametrize("meshlet_stream", [False, True], ids=["meshlet_buffer", "meshlet_stream"])
def test_frame_sequence_through_host_api(torch_mod, oracle, meshlet_stream):
    torch = torch_mod
    W, H = 640, 360
    scene = sc.make_scene(41, 1500, n_meshes=200, extent=(50.0, 10.0, 50.0))
    cam = sc.default_camera()
    ctx = passes.Context(0)
    g = dict(draws=dev(torch, scene.entity_draw_buffer()), mesh=dev(torch, scene.mesh_infos),
             ent=dev(torch, scene.entities), mlt=dev(torch, scene.meshlets), mat=dev(torch, scene.materials))
    if meshlet_stream:
        # what GpuAssets::add_mesh does after uploading meshlets (Context::sync_meshlet_stream) — mesh by mesh into a
        # buffer that is then re-created larger at another address (everything uploaded so far must follow
        # the move, not only the newest range) — and what add_material does (the alpha classes)
        n = len(scene.meshlets)
        small = dev(torch, scene.meshlets[:n // 2])
        ctx.sync_meshlet_stream(small, 0, n // 4, capacity=n // 2)
        ctx.sync_meshlet_stream(small, n // 4, n // 2 - n // 4, capacity=n // 2)
        ctx.sync_meshlet_stream(g["mlt"], n // 2, n - n // 2)  # the grown buffer: [0, n/2) re-derived from it too
        ctx.sync_meshlet_stream_materials(g["mat"], len(scene.materials))
        del small
    rng = np.random.default_rng(3)
    evis0 = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis0 = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    evis, mvis = dev(torch, evis0), dev(torch, mvis0)
    assets = passes.Assets(g["mlt"], g["mesh"], g["mat"], material_count=len(scene.materials))
    hscene = passes.Scene(scene.entity_draw_count, g["draws"], g["ent"], meshlet_visibility_buffer=mvis)
    view = sc.mat4_cols(cam.view)
    proj = passes.Projection.Perspective(cam.fov, cam.z_near)
    planes = passes.frustum_planes_from_matrix(passes.projection_compute_matrix(proj, cam.aspect))[:5]  # forward.rs:264

    # -- EARLY: OcclusionCullInfo::VisibilityRead (forward.rs:266-298)
    ci1 = passes.CullInfo(view, planes, proj, occlusion_kind=passes.VISIBILITY_READ, visibility_buffer=evis,
                          meshlet_visibility_buffer=mvis)
    gpu1, disp = passes.create_meshlet_dispatch_command(ctx, "early_forward_depth_prepass", assets, hscene, ci1)
    draws = passes.create_meshlet_draw_commands(ctx, "early_forward_depth_prepass", assets, hscene, ci1, disp)
    torch.cuda.synchronize()
    assert gpu1["occlusion_pass"] == 1
    cap_d = (disp.size - 12) // 16
    od, _, _ = oracle.entity_cull(gpu1, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                  scene.entities, cap_d, evis0)
    nrec = int(od[:4].view(np.uint32)[0])
    assert nrec > 0 and np.array_equal(fetch(disp, 12 + 16 * nrec), od[:12 + 16 * nrec])
    odr, _, _ = oracle.meshlet_cull(gpu1, od, scene.meshlets, (draws.size - 4) // 28, scene.entities, scene.materials,
                                    mvis0)
    n1 = int(odr[:4].view(np.uint32)[0])
    assert n1 > 0 and np.array_equal(fetch(draws, 4 + 28 * n1), odr[:4 + 28 * n1])

    # the same cull through create_draw_commands (draw_gen.rs:239-322: both stages, as the shadow passes call it) — ONE
    # library call, which runs a scene of this size as one launch: same transient buffers, same bytes
    draws_b = passes.create_draw_commands(ctx, "early_forward_depth_prepass", assets, hscene, ci1)
    torch.cuda.synchronize()
    assert draws_b.ptr == draws.ptr and np.array_equal(fetch(draws_b, 4 + 28 * n1), odr[:4 + 28 * n1])
    assert np.array_equal(fetch(disp, 12 + 16 * nrec), od[:12 + 16 * nrec])

    # -- DepthPyramid::update (forward.rs:362-367)
    depth = sc.make_depth(9, W, H, cam)
    gdepth = dev(torch, depth)
    pyr_ptr, (w0, h0, mips) = passes.depth_pyramid_update(ctx, "depth_pyramid", gdepth, W, H)
    torch.cuda.synchronize()
    opyr, d = oracle.depth_reduce(depth, W, H)
    assert (w0, h0, mips) == (d.width, d.height, d.mip_levels)
    gp = passes.buffer_tensor(passes.HostBuffer(pyr_ptr, 4 * d.total_texels)).cpu().numpy().view(np.uint32)
    assert np.array_equal(gp, opyr.view(np.uint32))

    # -- LATE: OcclusionCullInfo::VisibilityWrite (forward.rs:371-403)
    ci2 = passes.CullInfo(view, planes, proj, occlusion_kind=passes.VISIBILITY_WRITE, visibility_buffer=evis,
                          meshlet_visibility_buffer=mvis, depth_pyramid=pyr_ptr, depth_pyramid_size=(w0, h0),
                          aspect_ratio=cam.aspect)
    gpu2, disp2 = passes.create_meshlet_dispatch_command(ctx, "late_forward_depth_prepass", assets, hscene, ci2)
    draws2 = passes.create_meshlet_draw_commands(ctx, "late_forward_depth_prepass", assets, hscene, ci2, disp2)
    torch.cuda.synchronize()
    od2, oevis, _ = oracle.entity_cull(gpu2, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                       scene.entities, cap_d, evis0, opyr, (w0, h0))
    nrec2 = int(od2[:4].view(np.uint32)[0])
    assert np.array_equal(fetch(disp2, 12 + 16 * nrec2), od2[:12 + 16 * nrec2])
    odr2, omvis, _ = oracle.meshlet_cull(gpu2, od2, scene.meshlets, (draws2.size - 4) // 28, scene.entities,
                                         scene.materials, mvis0, opyr, (w0, h0))
    n2 = int(odr2[:4].view(np.uint32)[0])
    assert n2 > 0 and np.array_equal(fetch(draws2, 4 + 28 * n2), odr2[:4 + 28 * n2])
    assert np.array_equal(evis.cpu().numpy().view(np.uint32), oevis)
    assert np.array_equal(mvis.cpu().numpy().view(np.uint32), omvis)

    # -- compute_clusters (app.rs:1204 -> cluster.rs:368)
    lights = sc.make_lights(5, 300)
    glights = dev(torch, lights)
    settings = passes.ClusterSettings(screen_resolution=(W, H))
    hscene_l = passes.Scene(scene.entity_draw_count, g["draws"], g["ent"], light_data_buffer=glights,
                            light_count=len(lights))
    out = passes.compute_clusters(ctx, settings, view, proj, cam.aspect, gdepth, (W, H), 1, hscene_l)
    torch.cuda.synchronize()
    info = out.info
    cc = list(info.cluster_count)
    push = np.zeros((), dtype=L.MARK_ACTIVE_PUSH)
    push["cluster_count"], push["tile_size_px"], push["screen_size"] = cc, info.tile_size_px, (W, H)
    push["z_near"], push["z_far"], push["z_scale"], push["z_bias"] = cam.z_near, 200.0, info.z_scale, info.z_bias
    push["depth_buffer_sample_count"] = 1
    om, ob = oracle.cluster_mark(push, depth)
    assert np.array_equal(fetch(out.tile_depth_slice_mask, om.nbytes).view(np.uint32), om)
    assert np.array_equal(fetch(out.depth_bounds, ob.nbytes).view(np.uint32).reshape(-1, 2), ob)
    ou, _ = oracle.cluster_compact(cc, om, (out.unique_cluster_buffer.size - 16) // 4)
    na = int(ou[12:16].view(np.uint32)[0])
    assert na > 0 and np.array_equal(fetch(out.unique_cluster_buffer, 16 + 4 * na), ou[:16 + 4 * na])
    cinfo = np.zeros((), dtype=L.CLUSTER_CULL_INFO)
    cinfo["world_to_view_matrix"] = view
    cinfo["screen_to_view_matrix"] = passes.mat4_inverse(passes.projection_compute_matrix(proj, cam.aspect))
    cinfo["cluster_count"], cinfo["tile_size_px"], cinfo["screen_size"] = cc, info.tile_size_px, (W, H)
    cinfo["z_near"], cinfo["z_far"], cinfo["global_light_count"] = cam.z_near, 200.0, len(lights)
    ol, oimg, _ = oracle.cluster_assign(cinfo, ou, ob, lights, (out.light_index_list.size - 4) // 4,
                                        cc[0] * cc[1] * cc[2])
    nl = int(ol[:4].view(np.uint32)[0])
    assert nl > 0 and np.array_equal(fetch(out.light_index_list, 4 + 4 * nl), ol[:4 + 4 * nl])
    gimg = fetch(out.light_offset_image, oimg.nbytes).view(np.uint32).reshape(-1, 2)
    active = ou[16:16 + 4 * na].view(np.uint32)
    assert np.array_equal(gimg[active], oimg[active])  # only active clusters are written (light_culling.comp:150)
    ctx.close()


def test_panics_like_the_reference(torch_mod):
    torch = torch_mod
    scene = sc.make_scene(42, 20)
    cam = sc.default_camera()
    ctx = passes.Context(0)
    g = [dev(torch, a) for a in (scene.entity_draw_buffer(), scene.mesh_infos, scene.entities, scene.meshlets,
                                 scene.materials)]
    assets = passes.Assets(g[3], g[1], g[4])
    hscene = passes.Scene(scene.entity_draw_count, g[0], g[2])
    proj = passes.Projection.Perspective(cam.fov, cam.z_near)
    too_many = np.tile(cam.planes[:1], (13, 1))
    with pytest.raises(passes.Panic, match="MAX_CULL_PLANES"):  # assert!, draw_gen.rs:334
        passes.create_meshlet_dispatch_command(ctx, "x", assets, hscene,
                                               passes.CullInfo(sc.mat4_cols(cam.view), too_many, proj))
    with pytest.raises(passes.Panic):  # VisibilityRead without a buffer: .unwrap(), draw_gen.rs:125-128
        passes.create_meshlet_dispatch_command(
            ctx, "y", assets, hscene,
            passes.CullInfo(sc.mat4_cols(cam.view), cam.planes, proj, occlusion_kind=passes.VISIBILITY_READ))
    ctx.close()
