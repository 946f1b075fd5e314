import json, sys
j=json.load(open(sys.argv[1]))
print('ms/step', round(j['ms_per_step'],3), 'value', round(j['value'],1))
for k in ('roofline','roofline_costreg','roofline_costvol','roofline_softmax','roofline_feature','cpu_baseline'):
    if k in j: print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in j[k].items() if a not in('kernel','sample')})
s=j['stage_ms_per_step']
print('feature',s['feature'], [(k.split('/')[1], v) for k,v in s.items() if k.startswith('feature/')])
for l in (2,1,0):
    print(l, 'hyp',s[f'hypotheses_{l}'],'cv',s[f'costvol_{l}'],'sm',s[f'softmax_{l}'], 'costreg', round(sum(v for k,v in s.items() if k.startswith(f'costreg_{l}/')),3), [ (k.split('/')[1], v) for k,v in s.items() if k.startswith(f'costreg_{l}/')])
