# round 6: instruction mix of rayleigh_forward<4, false>'s innermost loop (8 source-point pairs per lane and iteration: PPL 4 x unroll 2) -- the
# denominator of next_rows.rayleigh_forward.valu_frac in bench.py. Runs anywhere (hipcc cross-compiles). usage: bash scripts/r6/rayleigh_isa_counts.sh
cd "$(dirname "$0")/../.." && T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fgpu-flush-denormals-to-zero -fno-fast-math -S --cuda-device-only -Iinclude -o $T/r.s babelbrain_amd/csrc/bfd_rayleigh.hip 2>/dev/null
python3 - $T/r.s <<'PY'
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
i0 = [i for i, l in enumerate(src) if re.match(r'^_ZN\S*rayleigh_forwardILi4ELb0E\S*:', l)][0]
end = next(i for i in range(i0, len(src)) if '.amdhsa_kernel' in src[i])
f = src[i0:end]
h = next(i for i, l in enumerate(f) if 'Inner Loop Header: Depth=3' in l)
body = []
for l in f[h + 1:]:
    t = l.split(';')[0].strip()
    if t and not t.startswith('.'):
        body.append(t.split()[0])
        if t.startswith('s_cbranch'):
            break
c = collections.Counter(body)
cls = {'f64 arithmetic (4 cycles per wave64: 16 lanes per clock)': [k for k in c if k.endswith('_f64') or k.startswith('v_fmac_f64') or k.startswith('v_fract_f64')],
       'f64 <-> f32 conversions (4)': [k for k in c if k.startswith('v_cvt_')],
       'packed f32 (4: two values per lane)': [k for k in c if k.startswith('v_pk_')],
       'transcendental f32: v_sin, v_cos, v_rsq (8: quarter rate)': [k for k in c if k.split('_')[1] in ('sin', 'cos', 'rsq', 'exp', 'rcp', 'sqrt')],
       'moves (2)': [k for k in c if k.startswith('v_mov')]}
cyc = {'f64 arithmetic (4 cycles per wave64: 16 lanes per clock)': 4, 'f64 <-> f32 conversions (4)': 4, 'packed f32 (4: two values per lane)': 4,
       'transcendental f32: v_sin, v_cos, v_rsq (8: quarter rate)': 8, 'moves (2)': 2}
tot = 0
seen = set()
for name, ks in cls.items():
    ks = [k for k in ks if k not in seen and k.startswith('v_')]
    seen.update(ks)
    n = sum(c[k] for k in ks)
    tot += n * cyc[name]
    print('%4d  %s   %s' % (n, name, ' '.join('%s x%d' % (k, c[k]) for k in sorted(ks))))
rest = [k for k in c if k.startswith('v_') and k not in seen]
print('other vector instructions:', {k: c[k] for k in rest})
print('all instructions of the loop: %d (vector %d); SIMD cycles per iteration %d = %.4f per pair and lane (8 pairs per lane, 64 lanes)' % (len(body), sum(v for k, v in c.items() if k.startswith('v_')), tot, tot / 8 / 64))
print('=> ceiling %.0f Gpairs/s at 1024 SIMDs x 2.4 GHz' % (1024 * 2.4e9 / (tot / 8 / 64) / 1e9))
PY
rm -rf $T
