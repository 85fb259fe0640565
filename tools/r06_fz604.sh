#!/bin/bash
# the whole-mapper fuzz of seed 604 (SAM, switch variants), its first ITERS iterations, N times, optionally under switches:
# usage tools/r06_fz604.sh <label> <N> <ITERS> [VAR=value ...]
L=$1; N=$2; IT=$3; shift; shift; shift
for v in "$@"; do export "$v"; done
for i in $(seq 1 $N); do python tools/fuzz_mapper_gpu.py --iters $IT --reads 2000 --seed 604 --sam --variants 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', $i, d['bad_plain'], d['bad_extend'], {k:v['bad'] for k,v in d['by_variant'].items() if v['bad']}, [(e.get('it'), e.get('variant'), e.get('mode'), (e.get('only_library') or [''])[0][:40]) for e in d['examples'] if 'only_library' in e][:3])"; done
