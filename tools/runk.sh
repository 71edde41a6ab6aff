for k in 1 100 1000 10000; do RQ_SCAN_STATS=1 python bench.py --workload pq --k $k --no-cpu --no-host --steps 5 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('scan_stats'):
        s=json.loads(l[11:]); tot=sum(s[k] for k in ('lut','sample','stream','final_cut','sort_write')); print({k:round(100*s[k]/tot,1) for k in ('stream','final_cut','sort_write')}, 'alive1=%.4f' % (s['first_block_pushed']/s['first_block_rows']), s['n_items_filtered'], s['n_fallbacks'])
    elif l.startswith('{'): d=json.loads(l); print('K', d['config']['k'], d['ms_per_step'], d['checks'].get('gpu_vs_reference_bit_exact', d['checks']))
"; done
