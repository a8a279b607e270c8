#!/usr/bin/env python
"""profiles/<round>_*_pmc_{FETCH,WRITE}_SIZE.txt (tools/rocpd_pmc.py output of separate rocprofv3 --pmc passes)
-> profiles/<round>_pmc_traffic.json, the per-launch HBM/fabric bytes bench.py quotes as `roofline.traffic`.

  python tools/make_pmc_traffic.py r02 [dir = profiles]

traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 for wide coalesced streams (gfx950: FETCH_SIZE reports half
of a 16-B/lane stream, MI355X_MICROARCH.md); the conv kernels gather 64-byte segments, for which the counter is
NOT halved, so theirs is (FETCH_SIZE + WRITE_SIZE) * 1024 with the doubled figure kept as traffic_bytes_upper."""
import json
import os
import sys

PIXELS = {  # bench.py's conv names -> a substring of the kernel symbol
    "conv1_fwd": "conv1_fwd_patch_kernel", "conv2_fwd": "conv_patch_kernel<1, 2, true",
    "conv3_fwd": "conv_patch16_kernel<2, 3, true", "conv2_dgrad": "conv_patch_kernel<2, 1, false",
    "conv3_dgrad": "conv_patch16_kernel<3, 2, false", "conv1_wgrad": "conv1_wgrad_roles_kernel",
    "conv2_wgrad": "conv_wgrad_tr2_kernel<32", "conv3_wgrad": "conv_wgrad_tr2_kernel<64"}
RECURRENT = {"rnn_fwd_step_kernel": "rnn_fwd_step_kernel", "rnn_bwd_step_kernel": "rnn_bwd_step_kernel",
             "sgemm_grouped_kernel": "sgemm_grouped_kernel", "xgemm_kernel": "xgemm_kernel"}
# the cluster recurrence's instantiations (lr_rnn_cluster.hip): bench.py's name "rnnc_fwd_kernel<G,CC>"
for _g, _cc in [(3, c) for c in range(1, 28)] + [(4, c) for c in range(1, 25)]:
  for _w in ("fwd", "bwd"):
    # (the profiler leaves some of these names mangled: both spellings)
    # (32-unit members; third template argument = units per member)
    # (round 6: a fourth template argument = samples per cluster; the B = 32 passes the lines profile are the 8-sample form)
    RECURRENT["rnnc_%s_kernel<%d,%d>" % (_w, _g, _cc)] = ("rnnc_%s_kernel<%d, %d, 32, 8>" % (_w, _g, _cc),
                                                          "rnnc_%s_kernelILi%dELi%dELi32ELi8E" % (_w, _g, _cc),
                                                          "rnnc_%s_kernel<%d, %d, 32>" % (_w, _g, _cc),
                                                          "rnnc_%s_kernelILi%dELi%dELi32EE" % (_w, _g, _cc))
MODELS = ("gru256", "lstm768", "lstm700", "lstm512", "gru800")


def read(path):
  rows = []
  if not os.path.exists(path):
    return rows
  for line in open(path):
    if line.startswith("#") or line.startswith("kernel "):
      continue
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES"):
      i = line.find(" " + counter + " ")
      if i > 0:
        vals = line[i + len(counter) + 2:].split()
        rows.append((line[:i].strip(), counter, int(vals[0]), float(vals[1])))
  return rows


def read_stats(path):
  """<round>_<model>_kernel_stats.txt (tools/rocpd_summary.py): kernel name -> average duration in us"""
  out = []
  if not os.path.exists(path):
    return out
  for line in open(path):
    if line.startswith("#") or line.startswith("kernel "):
      continue
    parts = line.rstrip().rsplit(None, 6)
    if len(parts) == 7:
      try:
        out.append((parts[0].strip(), float(parts[3])))
      except ValueError:
        pass
  return out


SIMDS = 256 * 4            # compute units x SIMDs
PEAK_CLOCK_MHZ = 2400.0    # cycles per microsecond at the peak engine clock (MI355X_MICROARCH.md)


def lookup(rows, sub, counter=None):
  subs = sub if isinstance(sub, tuple) else (sub,)
  for name, c, calls, avg in rows:
    if any(x in name for x in subs) and (counter is None or c == counter):
      return avg, calls, name
  return None


def main(tag, d="profiles"):
  out = {"_note": "HBM/fabric bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (bench.py "
                  "--no-graph, a few steps), averaged over the launches; see tools/make_pmc_traffic.py for the formula.",
         "source_files": []}
  # which kernels the passes ran: the fingerprint of every csrc/*.hip, header and compiler flag (lipreading_amd/_build.py),
  # written on the GPU box beside the passes (tools/refresh_profiles.sh).  bench.py quotes these figures only on a
  # build with the same fingerprint.
  fp = os.path.join(d, "%s_source_fingerprint.txt" % tag)
  out["source_fingerprint"] = open(fp).read().strip() if os.path.exists(fp) else None
  try:
    import subprocess
    out["git_head_when_assembled"] = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
  except Exception:
    out["git_head_when_assembled"] = None
  for model, table, doubled in [(m, RECURRENT, True) for m in MODELS] + [("pixels", PIXELS, False)]:
    f = read(os.path.join(d, "%s_%s_pmc_FETCH_SIZE.txt" % (tag, model)))
    w = read(os.path.join(d, "%s_%s_pmc_WRITE_SIZE.txt" % (tag, model)))
    sq = read(os.path.join(d, "%s_%s_pmc_SQ_pass1.txt" % (tag, model)))
    stats = read_stats(os.path.join(d, "%s_%s_kernel_stats.txt" % (tag, model)))
    if not f or not w:
      continue
    out["source_files"] += ["%s/%s_%s_pmc_%s.txt" % (d, tag, model, c) for c in ("FETCH_SIZE", "WRITE_SIZE")]
    sec = {}
    for key, sub in table.items():
      a, b = lookup(f, sub), lookup(w, sub)
      if a is None or b is None:
        continue
      rec = {"kernel": a[2], "calls": a[1], "FETCH_SIZE_KiB_avg": round(a[0], 1), "WRITE_SIZE_KiB_avg": round(b[0], 1)}
      if doubled:
        rec["traffic_bytes"] = int((2 * a[0] + b[0]) * 1024)
      else:
        rec["traffic_bytes"] = int((a[0] + b[0]) * 1024)
        rec["traffic_bytes_upper"] = int((2 * a[0] + b[0]) * 1024)
      busy, wave = lookup(sq, sub, "SQ_VALU_MFMA_BUSY_CYCLES"), lookup(sq, sub, "SQ_WAVE_CYCLES")
      wait = lookup(sq, sub, "SQ_WAIT_ANY")
      if busy and wave and wave[0] > 0:
        # Two normalisations.  mfma_busy: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x the kernel's average duration x the
        # 2.4 GHz peak clock) — the fraction of the chip's matrix-pipe cycles the launch used, whatever its occupancy
        # (a lower bound where the clock sagged under the profiler).  mfma_busy_per_wave_cycle: / (4 x SQ_WAVE_CYCLES),
        # rounds 1-4's figure: the matrix pipe's share of the cycles a wave was RESIDENT, right only for kernels
        # with one wave per SIMD (round 4 quoted 0.25-0.36 for layer 3's two-workgroups-per-CU kernels, which run at
        # 0.50 of the flop peak: each of their waves' cycles was counted, the pipe they share only once).
        subs = sub if isinstance(sub, tuple) else (sub,)
        dur = next((us for name, us in stats if any(x in name for x in subs)), None)
        rec["mfma_busy_per_wave_cycle"] = round(busy[0] / (4.0 * wave[0]), 4)
        if dur:
          rec["avg_us"] = dur
          rec["mfma_busy"] = round(busy[0] / (SIMDS * dur * PEAK_CLOCK_MHZ), 4)
        if wait:
          rec["wait_any_frac"] = round(wait[0] / wave[0], 4)
      sec[key] = rec
    out[model] = sec
  path = os.path.join(d, "%s_pmc_traffic.json" % tag)
  with open(path, "w") as fh:
    json.dump(out, fh, indent=1)
  print(path, {k: sorted(v) for k, v in out.items() if isinstance(v, dict)})


if __name__ == "__main__":
  main(*sys.argv[1:])
