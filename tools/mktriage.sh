# usage: bash tools/mktriage.sh      (run here)  ->  tools/variants/triage.so
# The product library carries no triage code: the ORBIT_SP_DEBUG switches, the per-phase cycle stamps and the per-wave
# wall-clock stamps of meshlet_eval exist only in -DORBIT_TRIAGE builds of meshlet_eval.hip and abi.hip, the
# executed-test counters of the cluster assignment only in such a build of light_cluster.hip.  This builds such a
# library next to the product (which is not touched); tools/phase.py, wave_tail.py and ab_inproc.py load it
# (tools/_triage.py) and refuse to run without it.
exec bash "$(dirname "$0")/mkvariant_any.sh" triage meshlet_eval,meshlet_emit,abi,light_cluster -DORBIT_TRIAGE "$@"
