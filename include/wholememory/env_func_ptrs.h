/* forwarding header: lets a binding written against libwholegraph (#include <wholememory/env_func_ptrs.h>) compile against libwholegraph_amd */
#include "../wholegraph_amd.h"
