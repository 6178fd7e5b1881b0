// capi_internal.h -- definitions of the opaque handles of include/rt_hip.h shared by capi.cpp and render.hip.
#pragma once
#include <string>
#include "../../include/rt_hip.h"
#include "scene_builder.h"

struct RtBuilder { rt::Builder b; };
struct RtFlat { rt::FlatScene flat; RtSceneView view; };

extern thread_local std::string g_rt_error;
int rt_set_error(int code, const std::string& msg);
