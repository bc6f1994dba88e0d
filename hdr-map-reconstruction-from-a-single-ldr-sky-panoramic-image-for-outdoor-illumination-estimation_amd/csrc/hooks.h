// Every HDRSKY_* environment variable the library looks at, read ONCE into one structure (at the first launch, or again by
// hdrsky_hooks_reload() - the tests flip switches inside one process): no getenv() on a launch path.
//
// Two classes.  SWITCHES select between shipped code paths and are always honoured (the test-suite uses them to prove which
// kernel ran and to compare paths bit for bit).  TUNING HOOKS belong to the A/B experiments under profiles/ (tile shapes,
// workgroup budgets, split factors); they are honoured only when HDRSKY_EXPERIMENTS=1 is set and read as their defaults
// otherwise, so a stray variable in a deployment's environment cannot change what the product runs.
#pragma once

struct HdrskyTileHook { int set; int v[6]; };   // "wm,wn,mi,ni,tw[,db]"
// HDRSKY_TILE_RULES="cout_lo,cout_hi,m_lo,m_hi,cin_lo,cin_hi,kh,ph=wm,wn,mi,ni,tw,db;...": table entries injected for an A/B run
// (kh / ph: -1 = any).  The first matching rule wins, ahead of the table (single-product mode only).
struct HdrskyTileRule { long lo[3], hi[3]; int kh, ph; int v[6]; };

struct HdrskyHooks {
  // ---- switches ---------------------------------------------------------------------------------------------------
  int da_region;       // HDRSKY_DA_REGION   1: LDS-region kernels of the distortion-aware conv where they fit (default);
                       //                    0: never; 2: fail with HDRSKY_EUNSUPPORTED instead of falling back
  int da_tm;           // HDRSKY_DA_TM       pixels per tile of the region kernel: 0 = chosen per launch (default), 32, 64
  int da_tpr;          // HDRSKY_DA_TPR      upper limit of filter taps per barrier round (0 = none)
  int da_tab;          // HDRSKY_DA_TAB      0: no per-workgroup sample table in da_conv_kernel (default 1)
  int no_phase;        // HDRSKY_NO_PHASE    stride-2 data gradients on the zero-stuffed operand instead of by output phases
  int no_dot1;         // HDRSKY_NO_DOT1     the one-output-channel conv through the MFMA tile instead of conv_dot1_kernel
  int wgrad2;          // HDRSKY_WGRAD2      0: weight gradients never on conv_wgrad2_kernel (default 1)
  int wgrad3;          // HDRSKY_WGRAD3      0: ... never on conv_wgrad3_kernel (default 1)
  int tile_table;      // HDRSKY_TILE_TABLE  4: the tile table of rounds 1-4 (tuned on launch latency alone); default 5: re-measured saturated
  int nab_one;         // HDRSKY_NAB_ONE     0: InstanceNorm backward never on the one-launch register-resident kernel (default 1)
  // ---- tuning hooks (HDRSKY_EXPERIMENTS=1) --------------------------------------------------------------------------
  int experiments;
  int ntile_rules; HdrskyTileRule tile_rules[16];
  HdrskyTileHook tile, tile_t16, tile_wide, tile_c32, tile_c16, tile_c64;   // HDRSKY_TILE, _T16, _WIDE, _C32, _C16, _C64 (64->64 from 65536 pixels)
  int wgrad2_s2min;    // HDRSKY_WGRAD2_S2MIN   (32)
  int wgrad2_mint;     // HDRSKY_WGRAD2_MINT    (2)
  int wgrad2_wgs;      // HDRSKY_WGRAD2_WGS     (0 = 192 workgroups shared out by work)
  int wgrad3_minpx;    // HDRSKY_WGRAD3_MINPX   (256)
  int wgrad3_wgs;      // HDRSKY_WGRAD3_WGS     (256)
  int wgrad_set, wgrad[3];   // HDRSKY_WGRAD "workgroups,force_small,workgroups_grouped"
  int da_group;        // HDRSKY_DA_GROUP       (-1)
  int da_wg_group;     // HDRSKY_DA_WG_GROUP    (-1)
  int fc_nsplit;       // HDRSKY_FC_NSPLIT      (4)
  int fc_update_nb;    // HDRSKY_FC_UPDATE_NB   (0: one 32x32 MFMA block per wave of the fused Dense update; 2: two, the default of rounds 3-4)
  int fc_update_rows;  // HDRSKY_FC_UPDATE_ROWS (0) grid rows of the fused Dense update (0 = one workgroup row per 32-k tile; fewer: the workgroups walk their tiles)
  int fc_rg;           // HDRSKY_FC_RG          (reduction groups per workgroup of fc_mfma_kernel: 1, 2 or 4)
  int nab_target;      // HDRSKY_NAB_TARGET     (512)
  int wgrad2_nt;       // HDRSKY_WGRAD2_NT      (0) non-temporal operand copies of conv_wgrad2_kernel: 1 x, 2 dy
  int nab_nt;          // HDRSKY_NAB_NT         (0) non-temporal loads of x in the one-launch InstanceNorm backward
  int conv_epi_lds;    // HDRSKY_CONV_EPI_LDS   (0) conv_igemm, direct-B: the epilogue tile gets LDS of its own while operand planes + tile stay within this many KB
                       //                       (rounds 3-4: 80, a finished wave writes its accumulators without waiting for the slowest; inside the step the
                       //                       smaller footprint - four workgroups of a 64-channel layer per compute unit instead of two - is worth more:
                       //                       step -0.8 %, forward -2.5 %, profiles/r05_conv_epi_lds_ab.txt)
  int fc_w_nt;         // HDRSKY_FC_W_NT        (1) non-temporal loads of the Dense layers' weight stream (fc_mfma_kernel)
  int opt_nt;          // HDRSKY_OPT_NT         (0) non-temporal stores of w / ms in rmsprop2_kernel
  int fc_nt;           // HDRSKY_FC_NT          (1) non-temporal policy of the fused Dense update: 1 image stores, 2 w / ms stores, 4 w / ms loads
};

const HdrskyHooks& hdrsky_hooks();
