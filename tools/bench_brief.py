"""One-screen summary of a bench.py JSON line (headline + secondary legs). usage: bench_brief.py <file>"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
def brief(d, name):
    r, k, mp, g = d.get("roofline", {}), d.get("roofline_k123", {}), d.get("message_passing", {}), d.get("gemm", {})
    print(f"{name}: {d['ms_per_step']:.3f} ms/step  {d['value'] / 1e6:.1f} M layer-edges/s | roofline {r.get('bound')} frac {r.get('frac', 0):.3f}"
          f" exec {r.get('executed_mfma_frac') or 0:.3f} | k123 {k.get('ms_per_step', 0):.3f} ms survey {k.get('frac_of_survey_roofline') or 0:.3f}"
          f" own {k.get('own_frac_of_hbm_peak') or 0:.3f} | mp {mp.get('ms_per_step', 0):.3f} ms hbm {mp.get('frac_of_hbm_peak') or 0:.3f} | gemm {g.get('ms_per_step', 0):.3f} ms")
if "ms_per_step" in d:
    brief(d, "headline")
for k, v in (d.get("secondary") or {k: v for k, v in d.items() if k.startswith("batch_cycle")}).items():
    if "error" in v:
        print(k, "ERROR", v["error"])
    elif "ms_per_step" not in v:                  # the loader-batch cycle leg
        print(f"{k}: steady {v['steady_state_ms_per_step']:.3f} ms/step, known class {v['amortised_ms_per_step_known_class']:.3f} "
              f"({v['amortised_over_steady_known_class']:.3f} x), arena load {v['arena_load_ms']:.2f} ms, capture {v['capture_ms']:.1f} ms, "
              f"capture per batch {v['recapture_every_batch']['amortised_over_steady']:.3f} x, assemble {v['assemble_ms']:.1f} ms, "
              f"whole loop pipelined {(v.get('pipelined_loop') or {}).get('over_steady') or 0:.3f} x")
    else:
        brief(v, k)
if len(sys.argv) > 2 and "message_passing" in d:
    for k, v in sorted(d["message_passing"]["per_kernel_ms"].items()):
        print(f"   {k:50s} {v * 1e3:8.1f} us")
    for k, v in d["gemm"]["per_shape"].items():
        print(f"   {k:50s} x{v[0]:.0f} {v[1] * 1e3:8.1f} us  {v[2]:6.1f} TF")
