/* forwarding header: lets a binding written against libwholegraph (#include <wholememory/wholememory_op.h>) compile against libwholegraph_amd */
#include "../wholegraph_amd.h"
