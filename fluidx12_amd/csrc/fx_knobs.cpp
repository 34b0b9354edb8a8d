// fx_knobs.cpp -- storage of the launcher switches (fx_knobs.h) and fx_set_knob / fx_knob_name of the C ABI
#include "fx_knobs.h"
#include "../../include/fluidx_hip.h"
#include <string.h>

namespace {
const char* const kNames[] = {
	"ADVECT_ALPHA", "ADVECT_BLOCK", "ADVECT_DEFER", "ADVECT_FAST", "ADVECT_LDS", "ADVECT_LDS_HALF", "ADVECT_TILE_ROWS", "ADVECT_ZCHUNK",
	"BLOCK_REMAP", "BLOCK_SHAPE", "COMM_PRIORITY",
#ifdef FX_LAB
	"DEBUG_NO_COPY",
#endif
	"FREEZE_DENSE_LEVELS", "FREEZE_DENSE_ONE", "FREEZE_FAST", "FREEZE_FUSE_DIV", "FREEZE_NT", "FREEZE_SHRINK", "FREEZE_STRIP4", "FREEZE_T", "FREEZE_WGS",
	"JACOBI2D_TILE", "JACOBI_BLOCK", "JACOBI_BLOCKG", "JACOBI_PREFER3", "JACOBI_PREFER4", "JACOBI_T", "LIGHT_FILL", "LIGHT_FILL_DIRTY", "LIGHT_RAY_NT", "LIGHT_RAY_WGS", "PROJECT_V4", "RCCL_ONE_COMM", "ROW_VW",
	"STRIP3H_PAIRS", "STRIP3_COOP", "STRIP3_NO512", "STRIP3_OFF", "STRIP3_ZCHUNK", "STRIP4X", "STRIP4X_MINP", "STRIP4X_ORDER", "STRIP4X_WGS", "STRIP4_OCTET", "STRIP4_ZCHUNK", "STRIP_GENERIC", "STRIP_R", "STRIP_REMAP", "STRIP_WGS", "STRIP_WIDE",
	"STRIP_ZCHUNK", "VIEW_ORDER", "VIEW_WGS", "XCD_REMAP" };
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
