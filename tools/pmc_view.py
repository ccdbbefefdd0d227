"""tools/pmc_view.py summary.txt [kernel substrings...]: the SQ ratios of the kernels of a tools/pmc_*.sh summary"""
import re
import sys
txt = open(sys.argv[1]).read()
want = sys.argv[2:] or ['sk_']
for b in re.split(r'\n(?=\S)', txt):
    name = b.split('\n')[0]
    if not any(x in name for x in want):
        continue
    d = {}
    for l in b.split('\n')[1:]:
        m = re.match(r'\s+(\S+)\s+mean/dispatch\s+([\d.]+)', l)
        if m:
            d[m.group(1)] = float(m.group(2))
    wc = d.get('SQ_WAVE_CYCLES', 1) or 1
    g = lambda k: d.get(k, 0)
    print(name[:50])
    print('   waves %.0f  wave_cycles %.3g  busy_cycles %.3g' % (g('SQ_WAVES'), wc, g('SQ_BUSY_CYCLES')))
    print('   WAIT_ANY %.2f  WAIT_INST_ANY %.2f  ACTIVE_INST_ANY %.2f  ACTIVE_VALU %.3f ACTIVE_LDS %.3f WAIT_INST_LDS %.3f' %
          tuple(g(k) / wc for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS')))
    print('   INSTS: VALU %.4g SALU %.4g LDS %.4g VMEM_RD %.3g VMEM_WR %.3g | LDS conflict/active %.2f  LDS_IDX_ACTIVE %.3g' %
          (g('SQ_INSTS_VALU'), g('SQ_INSTS_SALU'), g('SQ_INSTS_LDS'), g('SQ_INSTS_VMEM_RD'), g('SQ_INSTS_VMEM_WR'),
           g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1), g('SQ_LDS_IDX_ACTIVE')))
    if 'FETCH_SIZE' in d or 'WRITE_SIZE' in d:
        print('   FETCH_SIZE %.4g KB  WRITE_SIZE %.4g KB  GRBM_GUI_ACTIVE %.3g' % (g('FETCH_SIZE'), g('WRITE_SIZE'), g('GRBM_GUI_ACTIVE')))
