"""Pretty-prints a bench.py JSON line:  python tools/show_bench.py <file>
   python tools/show_bench.py --design-table <file>   the measured columns (ms / step, fraction) of DESIGN.md section 2's per-kernel table as markdown rows, from the
   line's stage times and roofline objects: what the hand-written table is diffed against after a new driver-style run"""
import json, sys


def design_table(j):
    s, rows = j["stage_ms_per_step"], []
    lv = lambda key: [round(s.get(f"{key}_{l}", 0.0), 3) for l in (2, 1, 0)]
    layer = lambda name: [round(s.get(f"costreg_{l}/{name}", 0.0), 3) for l in (2, 1, 0)]
    tot = lambda xs: round(sum(xs), 2)
    frac = lambda key, sub="frac": (j.get(key) or {}).get(sub)
    rows.append(("hypotheses", tot(lv("hypotheses")), lv("hypotheses"), "-"))
    rows.append(("fused homo_warp + aggregation", tot(lv("costvol")), lv("costvol"), f"{frac('roofline_costvol'):.3f} of HBM; per level { {k: round(v, 3) for k, v in sorted((frac('roofline_costvol', 'per_level_frac') or {}).items(), reverse=True)} }"))
    rows.append(("un-fused homo_warp op (reference signature)", round((j.get("roofline_homo_warp") or {}).get("avg_launch_ms", 0.0), 3), "per call",
                 f"{frac('roofline_homo_warp'):.3f} dirtied / {j.get('roofline_homo_warp_frac_hot', 0):.3f} hot"))
    rows.append(("CostRegNet.conv0", tot(layer("conv0")), layer("conv0"), f"{frac('roofline'):.3f} of HBM (algorithmic)"))
    for names in (("conv1", "conv3"), ("conv5",), ("conv2", "conv4", "conv6"), ("conv7",), ("conv9",)):
        rows.append((" / ".join(names), " / ".join(str(tot(layer(n))) for n in names), [layer(n) for n in names], ""))
    tail = [round(a + b, 3) for a, b in zip(layer("conv11"), layer("prob"))]
    rows.append(("conv11 + prob + regression", tot(tail), tail, f"{frac('roofline_prob_regress'):.3f} of HBM"))
    rows.append(("FeatureNet", round(s.get("feature", 0.0), 2), {k.split('/')[1]: v for k, v in s.items() if k.startswith('feature/')},
                 f"{frac('roofline_feature'):.3f} executed / {((j.get('roofline_feature') or {}).get('all_float32') or {}).get('frac', 0):.3f} all-float32"))
    rows.append(("whole CostRegNet", round((j.get("roofline_costreg") or {}).get("ms_per_step", 0.0), 2), "",
                 f"{frac('roofline_costreg'):.3f} executed / {j.get('roofline_costreg_frac_all_float32', 0):.3f} all-float32 measured"))
    print(f"| stage | ms / step (batch {j['config'].get('batch_per_forward')}) | per level 2 / 1 / 0 or per layer | fraction |\n|---|---|---|---|")
    for r in rows:
        print("| " + " | ".join(str(x) for x in r) + " |")
    print(f"\nheadline {j['value']:.1f} {j['unit']} ({j['ms_per_step']:.3f} ms per step); batch 1 {j.get('batch1', {}).get('value', 0):.1f}"
          f" (two streams {j.get('batch1', {}).get('two_streams', {}).get('value', 0):.1f}); two streams x half the batch {j.get('two_streams', {}).get('value', 0):.1f}")


if sys.argv[1] == "--design-table":
    design_table(json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]))
    sys.exit(0)
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
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
