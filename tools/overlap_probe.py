"""Builds and runs tools/probes/overlap_probe.hip: can a GELU epilogue hide under another tile's K loop when two 4-wave groups of a workgroup take
turns?  Prints microseconds per 256 x 256-tile equivalent (serial: one tile by 8 waves; ping-pong: two 256 x 128 tiles, one per group)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.abspath(__file__))
so = '/tmp/liboverlap_probe.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-ffp-contract=off', '-Wno-pass-failed'] + [a for a in sys.argv[1:] if a.startswith('-')] + [
                       os.path.join(ROOT, 'probes', 'overlap_probe.hip'), '-o', so])
print('flags:', ' '.join(sys.argv[1:]) or '(plain -O3: the SLP vectoriser packs the epilogue into v_pk_*_f32)')
lib = ctypes.CDLL(so)
tiles = int(os.environ.get('OP_TILES', 200))
ms = (ctypes.c_float * 8)()
for rep in range(2):
    assert lib.overlap_probe(tiles, ms) == 0
    for s, name in enumerate(('v_mfma_f32_16x16x32_bf16', 'v_mfma_f32_32x32x16_bf16')):
        us = [ms[4 * s + m] * 1e3 / tiles for m in range(4)]
        print(f'{name}: per tile  serial {us[0]:6.2f} us   ping-pong {us[1]:6.2f} us   K loops only {us[2]:6.2f} us   epilogues only {us[3]:6.2f} us'
              f'   (ping-pong / serial {us[1] / us[0]:.2f}; K + epilogue {us[2] + us[3]:6.2f})', flush=True)
ms4 = (ctypes.c_float * 2)()
for rep in range(2):
    assert lib.overlap_probe4(tiles, ms4) == 0
    print(f'one wave, two half-size accumulator sets (16x16x32): epilogue slices inside the other set\'s K loop {ms4[0] * 1e3 / tiles:6.2f} us per tile, '
          f'one after the other {ms4[1] * 1e3 / tiles:6.2f} us   (ratio {ms4[0] / ms4[1]:.2f})', flush=True)
