/* Same-named shim for the reference's include/scene.h: the declarations live in
 * path_tracer_api.h so the seven legacy headers cannot drift apart. */
#pragma once
#include "path_tracer_api.h"
