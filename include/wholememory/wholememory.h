/* forwarding header: lets a binding written against libwholegraph (#include <wholememory/wholememory.h>) compile against libwholegraph_amd */
#include "../wholegraph_amd.h"
