#!/usr/bin/env python3
"""glTF 2.0 (.glb / .gltf) -> `.orbitscene` (the cull path's input buffers, orbit_amd/orbitscene.py), through the
host mirror's asset ingestion (orbit_amd/host/orbit_gltf.*: load_gltf + the LOD chain of GpuAssets::add_mesh) and
SceneData::update_scene.  With a camera the dump carries the main view's GpuCullInfo as well, so that
tools/run_orbitscene.py can run it as it is.

    python tools/make_test_glb.py /tmp/scene.glb
    python tools/gltf_to_orbitscene.py /tmp/scene.glb /tmp/scene.orbitscene --camera 0 1 6
    python tools/run_orbitscene.py /tmp/scene.orbitscene          # on the GPU box: cull + oracle check
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from orbit_amd import camera, gltf, orbitscene

ap = argparse.ArgumentParser()
ap.add_argument("gltf")
ap.add_argument("out")
ap.add_argument("--camera", type=float, nargs=3, default=None, metavar=("X", "Y", "Z"),
                help="adds a cull_info section: the renderer's default camera (fov 90, near 0.01, 16:9) at X Y Z")
a = ap.parse_args()
loaded = gltf.load(a.gltf)
b = gltf.to_scene_buffers(loaded)
sections = {k: b[k] for k in ("entity_draws", "mesh_infos", "entities", "meshlets", "materials", "meshlet_data",
                              "vertex_positions")}
if a.camera is not None:
    sections["cull_info"] = np.frombuffer(camera.frame_cull_info(tuple(a.camera)).tobytes(), dtype=np.uint8)
orbitscene.write(a.out, **sections)
mi = loaded["mesh_infos"]
print(f"{a.out}: {len(mi)} meshes ({int(mi['lod_count'].sum())} LOD levels), {len(loaded['meshlets'])} meshlets, "
      f"{len(loaded['materials'])} materials, {len(loaded['entities'])} nodes -> "
      f"{int(b['entity_draws'][:4].view(np.uint32)[0])} entity draws, {os.path.getsize(a.out)} bytes")
