/*
 * ddcmi_md -- stand-alone driver with ddcMD's command line for this path:
 *     ddcmi_md [-o object.data] [-r restart] [-d datafile]
 * (masterFactory.c:12-22: -o object file, -r restart file).  Loads the deck,
 * sets up the plugin objects (plugin.c) and runs simulateMaster.
 */
#include "plugin.h"
#include <string.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
   const char *obj = "object.data", *restart = NULL, *data = "data", *extra = NULL;
   for (int i = 1; i < argc; i++)
   {
      if (strcmp(argv[i], "-o") == 0 && i + 1 < argc) obj = argv[++i];
      else if (strcmp(argv[i], "-r") == 0 && i + 1 < argc) restart = argv[++i];
      else if (strcmp(argv[i], "-d") == 0 && i + 1 < argc) data = argv[++i];
      else if (strcmp(argv[i], "-x") == 0 && i + 1 < argc) extra = argv[++i];      /* extra object text compiled after the files */
      else { fprintf(stderr, "usage: %s [-o object.data] [-r restart] [-d datafile] [-x 'name CLASS {...}']\n", argv[0]); return 2; }
   }
   char err[1024];
   SIMULATE *sim = simulate_init(obj, restart, extra, err, sizeof(err));
   if (!sim) { fprintf(stderr, "ddcmi_md: %s\n", err); return 1; }
   int rc = simulateMaster(sim, data);
   simulate_free(sim);
   return rc;
}
