import re,sys
t=open(sys.argv[1]).read()
for k in sys.argv[2:]:
    i=t.find('### bmh::'+k)
    if i<0: print('missing',k); continue
    sec=t[i:t.find('###',i+5)]
    d={}
    for m in re.finditer(r'\| (\w+) \| ([\d.e+]+) \| (\d+) \|',sec): d[m.group(1)]=float(m.group(2))
    wc=d.get('SQ_WAVE_CYCLES',1)
    print(k, ' waves %.0f valu %.3g salu %.3g lds %.3g vmem_rd %.3g wr %.3g | active_any %.0f%% wait_inst %.0f%% wait_any %.0f%% of wave_cycles %.3g; busy %.3g'%(d.get('SQ_WAVES',0),d.get('SQ_INSTS_VALU',0),d.get('SQ_INSTS_SALU',0),d.get('SQ_INSTS_LDS',0),d.get('SQ_INSTS_VMEM_RD',0),d.get('SQ_INSTS_VMEM_WR',0),100*d.get('SQ_ACTIVE_INST_ANY',0)/wc,100*d.get('SQ_WAIT_INST_ANY',0)/wc,100*d.get('SQ_WAIT_ANY',0)/wc,wc,d.get('SQ_BUSY_CYCLES',0)))
