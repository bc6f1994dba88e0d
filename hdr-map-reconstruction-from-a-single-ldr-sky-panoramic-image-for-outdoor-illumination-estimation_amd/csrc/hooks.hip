// The library's environment switches, read once (hooks.h).  Host code only.
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "hooks.h"

namespace {

HdrskyHooks g_hooks;
std::once_flag g_once;
std::mutex g_mutex;

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

HdrskyTileHook env_tile(const char* name) {
  HdrskyTileHook t{};
  if (const char* e = getenv(name)) {
    t.v[5] = 0;
    t.set = sscanf(e, "%d,%d,%d,%d,%d,%d", &t.v[0], &t.v[1], &t.v[2], &t.v[3], &t.v[4], &t.v[5]) >= 5;
  }
  return t;
}

void read_hooks() {
  HdrskyHooks h{};
  h.da_region = env_int("HDRSKY_DA_REGION", 1);
  h.da_tm = env_int("HDRSKY_DA_TM", 0);
  h.da_tpr = env_int("HDRSKY_DA_TPR", 0);
  h.da_tab = env_int("HDRSKY_DA_TAB", 1);
  h.no_phase = getenv("HDRSKY_NO_PHASE") != nullptr;
  h.no_dot1 = getenv("HDRSKY_NO_DOT1") != nullptr;
  h.wgrad2 = env_int("HDRSKY_WGRAD2", 1) != 0;
  h.wgrad3 = env_int("HDRSKY_WGRAD3", 1) != 0;
  h.nab_one = env_int("HDRSKY_NAB_ONE", 1);
  h.tile_table = env_int("HDRSKY_TILE_TABLE", 5);
  h.experiments = env_int("HDRSKY_EXPERIMENTS", 0) == 1;
  // tuning hooks: their defaults unless the gate is open
  h.wgrad2_s2min = 32; h.wgrad2_mint = 2; h.wgrad2_wgs = 0; h.wgrad3_minpx = 256; h.wgrad3_wgs = 256;
  h.da_group = -1; h.da_wg_group = -1; h.fc_nsplit = 4; h.fc_update_nb = 0; h.fc_rg = 4; h.nab_target = 512; h.fc_nt = 1; h.fc_w_nt = 1; h.opt_nt = 0; h.wgrad2_nt = 0; h.nab_nt = 0; h.conv_epi_lds = 0;
  if (h.experiments) {
    h.tile = env_tile("HDRSKY_TILE"); h.tile_t16 = env_tile("HDRSKY_TILE_T16"); h.tile_wide = env_tile("HDRSKY_TILE_WIDE");
    if (const char* tr = getenv("HDRSKY_TILE_RULES")) {
      const char* p = tr;
      while (*p && h.ntile_rules < 16) {
        HdrskyTileRule r{};
        long q[8]; int v[6];
        if (sscanf(p, "%ld,%ld,%ld,%ld,%ld,%ld,%ld,%ld=%d,%d,%d,%d,%d,%d", &q[0], &q[1], &q[2], &q[3], &q[4], &q[5], &q[6], &q[7],
                   &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) == 14) {
          for (int i = 0; i < 3; ++i) { r.lo[i] = q[2 * i]; r.hi[i] = q[2 * i + 1]; }
          r.kh = (int)q[6]; r.ph = (int)q[7];
          for (int i = 0; i < 6; ++i) r.v[i] = v[i];
          h.tile_rules[h.ntile_rules++] = r;
        }
        while (*p && *p != ';') ++p;
        if (*p == ';') ++p;
      }
    }
    h.tile_c32 = env_tile("HDRSKY_TILE_C32"); h.tile_c16 = env_tile("HDRSKY_TILE_C16"); h.tile_c64 = env_tile("HDRSKY_TILE_C64");
    h.wgrad2_s2min = env_int("HDRSKY_WGRAD2_S2MIN", h.wgrad2_s2min);
    h.wgrad2_mint = env_int("HDRSKY_WGRAD2_MINT", h.wgrad2_mint);
    h.wgrad2_wgs = env_int("HDRSKY_WGRAD2_WGS", h.wgrad2_wgs);
    h.wgrad3_minpx = env_int("HDRSKY_WGRAD3_MINPX", h.wgrad3_minpx);
    h.wgrad3_wgs = env_int("HDRSKY_WGRAD3_WGS", h.wgrad3_wgs);
    if (const char* e = getenv("HDRSKY_WGRAD")) h.wgrad_set = sscanf(e, "%d,%d,%d", &h.wgrad[0], &h.wgrad[1], &h.wgrad[2]) >= 1;
    h.da_group = env_int("HDRSKY_DA_GROUP", h.da_group);
    h.da_wg_group = env_int("HDRSKY_DA_WG_GROUP", h.da_wg_group);
    h.fc_nsplit = env_int("HDRSKY_FC_NSPLIT", h.fc_nsplit);
    h.fc_update_nb = env_int("HDRSKY_FC_UPDATE_NB", h.fc_update_nb);
    h.fc_update_rows = env_int("HDRSKY_FC_UPDATE_ROWS", 0);
    h.fc_rg = env_int("HDRSKY_FC_RG", h.fc_rg);
    h.nab_target = env_int("HDRSKY_NAB_TARGET", h.nab_target);
    h.fc_nt = env_int("HDRSKY_FC_NT", h.fc_nt) & 7;
    h.fc_w_nt = env_int("HDRSKY_FC_W_NT", h.fc_w_nt) != 0;
    h.opt_nt = env_int("HDRSKY_OPT_NT", h.opt_nt) != 0;
    h.wgrad2_nt = env_int("HDRSKY_WGRAD2_NT", h.wgrad2_nt) & 3;
    h.nab_nt = env_int("HDRSKY_NAB_NT", h.nab_nt) != 0;
    h.conv_epi_lds = env_int("HDRSKY_CONV_EPI_LDS", h.conv_epi_lds);
  }
  std::lock_guard<std::mutex> lock(g_mutex);
  g_hooks = h;
}

}  // namespace

const HdrskyHooks& hdrsky_hooks() {
  std::call_once(g_once, read_hooks);
  return g_hooks;
}

extern "C" {

// [host] Reads the HDRSKY_* variables again (the test-suite changes them inside one process).  Not to be called while
// another thread is inside a launch function of the library.
int hdrsky_hooks_reload(void) {
  std::call_once(g_once, [] {});
  read_hooks();
  return HDRSKY_OK;
}

// [host] 1 when HDRSKY_EXPERIMENTS=1 opened the tuning hooks at the last (re)load
int hdrsky_experiments_enabled(void) { return hdrsky_hooks().experiments; }

}  // extern "C"
