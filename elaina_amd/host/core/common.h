// common.h -- shared host-side vocabulary (reference core/common.h, trimmed to what the
// hot path's host surface needs).
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <filesystem>
#include <string>

#include "util/json.h"

namespace elaina {

namespace fs = std::filesystem;
using std::string;

struct Vector2f { float x = 0, y = 0; };
struct Vector2i { int x = 0, y = 0; };
struct Vector3f { float x = 0, y = 0, z = 0; };
struct AABB2f { Vector2f min, max; };
struct AABB3f { Vector3f min, max; };

// reference core/common.h:235-241
enum class ExportImageChannel { DIRICHLET_SDF, NEUMANN_SDF, SOURCE, SOLUTION, CHANNEL_COUNT };
// reference util/tonemapping.cuh:6-13
enum class ToneMapping { NONE, NONE_NORMALIZED, MATLAB_JET, MATLAB_PARULA, IDL_RDBU };

bool parse_channel(const string &name, ExportImageChannel *out);
bool parse_tone(const string &name, ToneMapping *out);
const char *channel_name(ExportImageChannel c);

// reference core/logger.h levels, plain stderr output
enum class LogLevel { Debug, Info, Success, Warning, Error };
void log_message(LogLevel level, const char *fmt, ...);
#define ELAINA_LOG(level, ...) ::elaina::log_message(::elaina::LogLevel::level, __VA_ARGS__)

}  // namespace elaina
