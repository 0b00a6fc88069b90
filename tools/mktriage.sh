# usage: bash tools/mktriage.sh      (run here)  ->  tools/variants/triage.so
# The product library carries no triage code: the ORBIT_SP_DEBUG switches, the per-phase cycle stamps and the per-wave
# wall-clock stamps of meshlet_eval exist only in -DORBIT_TRIAGE builds of meshlet_cull.hip and abi.hip, the executed-test counters of the cluster
# assignment only in such a build of light_cluster.hip.  This builds
# such a library next to the product (which is not touched); tools/phase.py, wave_tail.py and ab_inproc.py load it
# (tools/_triage.py) and refuse to run without it.
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -DORBIT_TRIAGE"
make -C orbit_amd/csrc -s -j4 || exit 1
mkdir -p tools/variants
for u in meshlet_cull abi light_cluster; do
  /opt/rocm/bin/hipcc $flags "$@" -c orbit_amd/csrc/$u.hip -o tools/variants/triage_$u.o || exit 1
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/variants/triage.so tools/variants/triage_meshlet_cull.o \
    tools/variants/triage_abi.o tools/variants/triage_light_cluster.o orbit_amd/csrc/scan.o orbit_amd/csrc/entity_cull.o \
    orbit_amd/csrc/depth_reduce.o || exit 1
rm -f tools/variants/triage_*.o
echo "built tools/variants/triage.so"
