// fx_knobs.cpp -- storage of the launcher switches (fx_knobs.h) and fx_set_knob / fx_knob_name of the C ABI
#include "fx_knobs.h"
#include "../../include/fluidx_hip.h"
#include <string.h>

namespace {
// The switches a caller could need (which path serves a geometry; the transport's two options): what fx_knob_name enumerates in the
// shipped library.  Everything else -- the A/B switches of the kernel launchers that docs/LAB.md measured with: chunk sizes, tile
// orders, superseded kernels -- exists only in a lab build (-DFX_LAB: FLUIDX_BUILD_LAB=1 python -m fluidx12_amd.build); in the shipped
// library a launcher that asks for one of those gets its default (knob_slot = -1) and fx_set_knob refuses the name.
const char* const kNames[] = {
	"ADVECT_ALPHA", "ADVECT_DEFER", "ADVECT_LDS", "COMM_PRIORITY", "FREEZE_STRIP4", "JACOBI_PREFER3", "JACOBI_PREFER4", "JACOBI_T", "LIGHT_FILL", "RCCL_ONE_COMM",
#ifdef FX_LAB
	"ADVECT_BLOCK", "ADVECT_FAST", "ADVECT_LDS_HALF", "ADVECT_TILE_ROWS", "ADVECT_ZCHUNK",
	"BLOCK_REMAP", "BLOCK_SHAPE", "DEBUG_NO_COPY",
	"FREEZE_DENSE_LEVELS", "FREEZE_DENSE_ONE", "FREEZE_FAST", "FREEZE_FUSE_DIV", "FREEZE_NT", "FREEZE_SHRINK", "FREEZE_T", "FREEZE_WGS",
	"JACOBI2D_TILE", "JACOBI_BLOCK", "JACOBI_BLOCKG", "LIGHT_FILL_DIRTY", "LIGHT_RAY_NT", "LIGHT_RAY_WGS", "PROJECT_V4", "ROW_VW",
	"STRIP3H_PAIRS", "STRIP3_COOP", "STRIP3_NO512", "STRIP3_OFF", "STRIP3_ZCHUNK", "STRIP4T", "STRIP4T_256", "STRIP4T_512", "STRIP4T_FROM", "STRIP4T_GRID", "STRIP4T_NARROW", "STRIP4T_NARROW_FROM", "STRIP4T_PIECES", "STRIP4X", "STRIP4X_MINP", "STRIP4X_NT", "STRIP4X_ORDER", "STRIP4X_WGS", "STRIP4_OCTET", "STRIP4_ZCHUNK", "STRIP4_ZFLOOR", "STRIP_GENERIC", "STRIP_R", "STRIP_REMAP", "STRIP_WGS", "STRIP_WIDE",
	"STRIP_ZCHUNK", "VIEW_ORDER", "VIEW_WGS", "XCD_REMAP",
#endif
};
const int kCount = (int)(sizeof kNames / sizeof kNames[0]);
struct Slot { bool set; char value[48]; };
Slot g_slots[kCount];
}  // namespace

namespace fx {
int knob_slot(const char* name)
{
	for (int i = 0; i < kCount; ++i) if (!strcmp(kNames[i], name)) return i;
	return -1;
}
const char* knob_at(int slot) { return slot >= 0 && slot < kCount && g_slots[slot].set ? g_slots[slot].value : nullptr; }
}  // namespace fx

extern "C" {

int fx_set_knob(const char* name, const char* value)
{
	if (!name) return FX_E_INVALID;
	const int i = fx::knob_slot(name);
	if (i < 0) return FX_E_INVALID;
	if (!value) { g_slots[i].set = false; return FX_OK; }
	if (strlen(value) >= sizeof g_slots[i].value) return FX_E_INVALID;
	strcpy(g_slots[i].value, value);
	g_slots[i].set = true;
	return FX_OK;
}

const char* fx_knob_name(uint32_t index) { return index < (uint32_t)kCount ? kNames[index] : nullptr; }

}  // extern "C"
