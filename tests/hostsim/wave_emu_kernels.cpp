// wave_emu_kernels.cpp — TEST-ONLY: the wavefront-level LGSSM kernels of the product ((4,4,2): csrc/lgssm_m4.h over lgssm_q4.h;
// (16,16,2): csrc/lgssm_n16.h), the same bodies the __global__ functions of kvae_lgssm_n16.hip wrap, run on emulated wavefronts
// (wave_emu.h) over host pointers.  hostsim.cpp routes its smoother entry points here when kvae_hostsim_wave_emu(1) was called.
#define KVAE_HOSTSIM 1
#define KVAE_WAVE_EMU 1
#include "wave_emu.h"

#include "../../kalman-vae_amd/csrc/lgssm_m4.h"
#include "../../kalman-vae_amd/csrc/lgssm_n16.h"

using namespace kvae;

static int g_launches[4] = {0, 0, 0, 0};   // emulated launches so far: fwd n4, bwd n4, fwd n16, bwd n16 (tests assert they happened)

template <bool HAS_FP, bool HAS_GQ>
static void bwd_n4(const kvae_lgssm_problem *p, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                   const kvae_lgssm_input_grads *out, float *ws) {   // as launch_bwd_m4 of kvae_lgssm_n16.hip
  const unsigned grid = (unsigned)((p->B + 15) / 16);
  if (m4::kv_m4_split_bwd(*p)) {
    wemu::launch(grid, [&] { m4::smooth_bwd_wave<HAS_FP, HAS_GQ>(*p, *saved, *up, *out, ws, KV_M4_BWD_CHAIN); });
    wemu::launch(m4::kv_m4_gain_grid(*p), [&] { m4::rts_bwd_items<HAS_FP>(*p, *saved, *up, *out, ws); });
    wemu::launch(grid, [&] { m4::smooth_bwd_wave<HAS_FP, HAS_GQ>(*p, *saved, *up, *out, ws, KV_M4_BWD_FCHAIN); });
    wemu::launch(m4::kv_m4_item_grid(*p), [&] { m4::filter_bwd_items(*p, *saved, *out, ws); });
    return;
  }
  wemu::launch(grid, [&] { m4::smooth_bwd_wave<HAS_FP, HAS_GQ>(*p, *saved, *up, *out, ws, KV_M4_BWD_ALL); });
}
extern "C" {

int kvae_wemu_m4_split_max_b(int v) {   // < 0: back to the default; returns the previous override
  const int was = m4::kv_m4_split_override();
  m4::kv_m4_split_override() = v;
  return was;
}
int kvae_wemu_launches(int which) { return which >= 0 && which < 4 ? g_launches[which] : -1; }

void kvae_wemu_fwd_n4(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, int do_filter, int do_rts) {
  const unsigned grid = (unsigned)((p->B + 15) / 16);
  g_launches[0] += 1;
  if (m4::kv_m4_split(*p, do_filter, do_rts)) {   // as launch_fwd_m4 of kvae_lgssm_n16.hip: filter | all gains at once | smoother
    const unsigned gg = m4::kv_m4_gain_grid(*p);
    if (st->aux) {
      wemu::launch(grid, [&] { m4::smooth_fwd_wave<true>(*p, *st, 1, 0); });
      wemu::launch(gg, [&] { m4::gains_wave<true>(*p, *st); });
      wemu::launch(grid, [&] { m4::smooth_fwd_wave<true>(*p, *st, 0, KV_M4_RTS_WITH_GAINS); });
    } else {
      wemu::launch(grid, [&] { m4::smooth_fwd_wave<false>(*p, *st, 1, 0); });
      wemu::launch(gg, [&] { m4::gains_wave<false>(*p, *st); });
      wemu::launch(grid, [&] { m4::smooth_fwd_wave<false>(*p, *st, 0, KV_M4_RTS_WITH_GAINS); });
    }
    return;
  }
  if (st->aux) wemu::launch(grid, [&] { m4::smooth_fwd_wave<true>(*p, *st, do_filter, do_rts); });
  else wemu::launch(grid, [&] { m4::smooth_fwd_wave<false>(*p, *st, do_filter, do_rts); });
}
void kvae_wemu_bwd_n4(const kvae_lgssm_problem *p, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                      const kvae_lgssm_input_grads *out, float *ws, int has_fp) {
  const bool gq = out->gQ.ptr != nullptr;
  g_launches[1] += 1;
  if (has_fp && gq) bwd_n4<true, true>(p, saved, up, out, ws);
  else if (has_fp) bwd_n4<true, false>(p, saved, up, out, ws);
  else if (gq) bwd_n4<false, true>(p, saved, up, out, ws);
  else bwd_n4<false, false>(p, saved, up, out, ws);
}

void kvae_wemu_fwd_n16(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, int do_filter, int do_rts) {
  n16::Lds L;   // one wavefront at a time: the tile of the workgroup in flight
  memset(&L, 0xFF, sizeof(L));
  g_launches[2] += 1;
  if (st->aux) wemu::launch((unsigned)p->B, [&] { n16::smooth_fwd_wave<true>(*p, *st, do_filter, do_rts, L); });
  else wemu::launch((unsigned)p->B, [&] { n16::smooth_fwd_wave<false>(*p, *st, do_filter, do_rts, L); });
}
void kvae_wemu_bwd_n16(const kvae_lgssm_problem *p, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                       const kvae_lgssm_input_grads *out, float *ws, int has_fp) {
  n16::Lds L;
  memset(&L, 0xFF, sizeof(L));
  const bool gq = out->gQ.ptr != nullptr;
  g_launches[3] += 1;
  if (has_fp && gq) wemu::launch((unsigned)p->B, [&] { n16::smooth_bwd_wave<true, true>(*p, *saved, *up, *out, ws, L); });
  else if (has_fp) wemu::launch((unsigned)p->B, [&] { n16::smooth_bwd_wave<true, false>(*p, *saved, *up, *out, ws, L); });
  else if (gq) wemu::launch((unsigned)p->B, [&] { n16::smooth_bwd_wave<false, true>(*p, *saved, *up, *out, ws, L); });
  else wemu::launch((unsigned)p->B, [&] { n16::smooth_bwd_wave<false, false>(*p, *saved, *up, *out, ws, L); });
}
}
