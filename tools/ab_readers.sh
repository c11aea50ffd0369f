run() { echo "== $1"; shift; env "$@" timeout 300 python tools/scope_bench.py --skip-b --templates 64000000 --repeat-block --threads $T 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['E']; print(d['seconds'], d['M_templates_per_s'], d['M_templates_per_s_steady']); print(d['stages'][1])"; }
T=16 run "t16 old-like" FQTK_NO_COUNT_ASSISTANT=1
T=16 run "t16 assistant" A=1
T=32 run "t32 helpers1 noassist" FQTK_NO_COUNT_ASSISTANT=1 FQTK_COPY_HELPERS=1
T=32 run "t32 default (3 helpers + assistant)" A=1
T=32 run "t32 helpers 2 + assistant" FQTK_COPY_HELPERS=2
