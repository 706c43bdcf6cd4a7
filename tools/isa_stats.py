import sys,re,collections
lines=open(sys.argv[1]).read().split('\n')
key=sys.argv[2]
start=[i for i,l in enumerate(lines) if key in l and l.startswith('_Z') and ': ' in l][0]
end=[i for i,l in enumerate(lines[start:]) if l.strip().startswith('s_endpgm')][0]+start
body=lines[start:end]
blocks=[];cur=('entry',[])
for l in body:
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m:
        blocks.append(cur);cur=(m.group(1),[])
    elif l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
        cur[1].append(l.strip())
blocks.append(cur)
for name,ins in blocks:
    c=collections.Counter(i.split()[0] for i in ins)
    tot=len(ins)
    if tot<30: continue
    top=', '.join(f'{k}:{v}' for k,v in c.most_common(16))
    print(name,tot,top)
