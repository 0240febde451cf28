// csrc-side view of the public C ABI (include/aitk_mi355.h is the single source of truth).
#pragma once
#include "../../include/aitk_mi355.h"
