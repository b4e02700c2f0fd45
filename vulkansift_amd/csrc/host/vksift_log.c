/* vksift_log.c — coloured stdout logger with a global maximum level.
 * Same observable behaviour as the reference's vkenv logger (logger.c:55-84): "[vkenv:TAG]" prefix,
 * ANSI colour per level, messages above the global level are dropped; default level = ERROR+WARNING+INFO. */
#include "vksift_log.h"

#include <stdarg.h>
#include <stdio.h>

static vksift_log_level g_level = VKSIFT_LOGLVL_INFO;

void vksift_log_set_level(vksift_log_level lvl) { g_level = lvl; }
vksift_log_level vksift_log_get_level(void) { return g_level; }

void vksift_log(vksift_log_level lvl, const char *tag, const char *fmt, ...)
{
  static const char *colour[] = {"", "\033[1;31m", "\033[1;33m", "\033[0m", "\033[0;36m"};
  static const char *name[] = {"", "ERROR", "WARNING", "INFO", "DEBUG"};
  if (lvl == VKSIFT_LOGLVL_NONE || lvl > g_level)
    return;
  va_list ap;
  va_start(ap, fmt);
  fprintf(stdout, "%s[vkenv:%s][%s] ", colour[lvl], tag, name[lvl]);
  vfprintf(stdout, fmt, ap);
  fprintf(stdout, "\033[0m\n");
  fflush(stdout);
  va_end(ap);
}
