// Backbones built on the generic layer-graph engine (gnet.h): constructors only; every C-ABI entry point of ptta_api.hip
// forwards to the GNet methods.
#pragma once
#include "gnet.h"

GNet* nlspn_create(int n, int h, int w, const ptta_hparams* hp, int legacy_offset, int* rc);         // nlspn_api.hip
GNet* costdc_create(int n, int h, int w, const ptta_hparams* hp, float max_depth, int flags, int* rc);          // costdc_api.hip
