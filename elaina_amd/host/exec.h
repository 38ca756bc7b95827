// exec.h -- JSON-driven experiment entry point, same surface as the reference (exec.h:6).
#pragma once
#include "core/common.h"

void run_expr(elaina::fs::path conf_path);
