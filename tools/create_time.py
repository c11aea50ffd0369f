import time, sys
sys.path.insert(0, '/root/repo')
from fqtk_amd import BarcodeMatcher, synth
for k in (1,2,3,4,5):
    cfg = synth.CONFIGS[k]; b = synth.make_barcodes(cfg)
    BarcodeMatcher(b, cfg.max_mismatches, cfg.min_mismatch_delta)
    t=time.perf_counter(); m = BarcodeMatcher(b, cfg.max_mismatches, cfg.min_mismatch_delta); dt=time.perf_counter()-t
    print(k, 'create %.1f ms' % (dt*1e3), 'entries', m.memo_entries, 'kind', m.memo_kind)
