#!/bin/bash
# C4 bench line per (library variant, environment) pair:  tools/sweep_variants.sh "<lib-tag|-> VAR=val ..." ...
# A tag names findex_amd/lib/variants/libfmx_<tag>.so (tools/build_variant.sh); "-" is the product build.
for spec in "$@"; do
  set -- $spec; tag=$1; shift
  lib=""; [ "$tag" != "-" ] && lib="FMX_LIB=$PWD/findex_amd/lib/variants/libfmx_$tag.so"
  echo -n "$tag $* : "
  env $lib "$@" timeout -k 10 300 python bench.py --workload ${WL:-c4} --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('ms/step %.3f kernel %.3f value %.0f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))" || exit 1
done
