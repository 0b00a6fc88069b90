# usage: bash tools/mkvariant.sh <name> [extra hipcc flags...]   (run here; builds a library variant for tools/ab*.sh)
# The meshlet evaluation and emit units with the extra flags (the two that read the tuning macros of meshlet_common.h);
# see tools/mkvariant_any.sh.
name=$1; shift
exec bash "$(dirname "$0")/mkvariant_any.sh" "$name" meshlet_eval,meshlet_emit "$@"
