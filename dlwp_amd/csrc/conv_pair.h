// conv_pair.h -- two INDEPENDENT convolution launches of a training step in one kernel launch (r4, VERDICT r3 item 1: "more than one
// layer per launch" on grids under one round of workgroups).
//
// At 8 samples (one rank's share of a global batch of 64) a step of the config-3 U-Net is 28 launches of 5-42 us back to back, every
// one of them under one round of workgroups; a layer's weight gradient and its data gradient both read the layer's pre-activation
// gradient and nothing of each other.  On separate queues they do not overlap usefully (0.45-0.47 ms against
// 0.41 serial), inside ONE grid they do: the first blocks of the launch run the weight-gradient body, the others the data-gradient
// (= forward-family Winograd) body, and the hardware fills the CUs with whatever fits.  Same bodies, same arithmetic, same bits.
//
// Between dlwp_pair_begin and dlwp_pair_end the launchers of the two families hand their launch over instead of issuing it
// (dlwp_pair_stash_*: 1 = taken); dlwp_pair_end issues the pair as one launch when a fused instance is compiled for the two
// kernel instances (conv_pair.hip), one after the other otherwise.
#pragma once
#include <functional>
#include "common.h"

struct ConvArgs;
struct WgradArgs;

// The data gradient brackets the one launch that may be handed over (its in-place fast path: nothing of the same call follows it on
// the stream); any other forward-family launch inside a pair is issued at once.
void dlwp_pair_allow_fwd(dlwp_handle_t h, int on);
// forward family (conv_fwd.hip): the 8 x 32 / 4-wave / 32-channel Winograd instance, float32; variant 0 = 16 positions, 1 = the
// 9-position variant (WinoCfg::UPS: no fused instance, issued alone).  `launch` issues it alone.
int dlwp_pair_stash_fwd(dlwp_handle_t h, const ConvArgs& a, int variant, int grid, void (*launch)(const ConvArgs&, int, hipStream_t),
                        hipStream_t s);
// weight gradient (conv_bwd.hip): a channel-block Winograd instance WgCbCfg<th, tw, cib / 16, waves / (cib / 16), nt / cog>
int dlwp_pair_stash_wgrad(dlwp_handle_t h, const WgradArgs& a, int th, int tw, int waves, int nt, int cib, int grid,
                          void (*launch)(const WgradArgs&, int, hipStream_t), hipStream_t s);
// what the handed-over weight gradient's own call still has to issue BEHIND it (its slab sum, when no reductions are being deferred)
void dlwp_pair_after_wgrad(dlwp_handle_t h, std::function<int()> fn);
void dlwp_pair_free(dlwp_handle_t h);      // dlwp_destroy
