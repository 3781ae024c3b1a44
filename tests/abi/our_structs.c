/* the same table from the structs libddcmi's host layer is built with */
#include <stddef.h>
#include "plugin.h"
#include "layout_table.inc"
