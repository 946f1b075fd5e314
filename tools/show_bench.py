"""Pretty-prints a bench.py JSON line:  python tools/show_bench.py <file>"""
import json, sys
j = json.load(open(sys.argv[1]))
print('ms/step', round(j['ms_per_step'], 3), 'median', j.get('median_ms_per_step'), 'value', round(j['value'], 1), j['unit'])
for k in ('single_stream', 'batch1'):
    if k in j:
        print(k, round(j[k]['value'], 1), 'ms/step', round(j[k]['ms_per_step'], 3), 'concurrent', round(j[k].get('concurrent', {}).get('value', 0), 1))


def show(d, indent=''):
    for k in ('roofline', 'roofline_costreg', 'roofline_costvol', 'roofline_softmax', 'roofline_prob_regress', 'roofline_feature', 'roofline_homo_warp', 'cpu_baseline'):
        if k in d:
            print(indent + k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in d[k].items()
                               if a not in ('kernel', 'sample', 'traffic_note', 'traffic_source', 'per_level_frac') or (a == 'per_level_frac' and k == 'roofline_costvol')})
    s = d.get('stage_ms_per_step')
    if not s:
        return
    print(indent + 'feature', s.get('feature'), [(k.split('/')[1], v) for k, v in s.items() if k.startswith('feature/')])
    for l in (2, 1, 0):
        print(indent + str(l), 'hyp', s[f'hypotheses_{l}'], 'cv', s[f'costvol_{l}'], 'sm', s.get(f'softmax_{l}'), 'costreg',
              round(sum(v for k, v in s.items() if k.startswith(f'costreg_{l}/')), 3), [(k.split('/')[1], v) for k, v in s.items() if k.startswith(f'costreg_{l}/')])


show(j)
if 'batch1' in j and 'stage_ms_per_step' in j['batch1']:
    print('--- batch 1')
    show(j['batch1'], '  ')
