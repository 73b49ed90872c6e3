/* bh/bh_ini.h -- INI reader with the reference's types and entry points (bh_ini_parser, _create, _destroy):
 * "[name]" opens a section, "key=value" adds a key to the open section, blanks are removed everywhere in a
 * line first, lines starting with '#', ';', '!' or empty are skipped. Own implementation for link closure of
 * unchanged consumers (src/cli/bcnn_cl.c reads the [net] section through it). */
#ifndef BH_INI_H
#define BH_INI_H
#include "bh_string.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct {
    char *name;
    char *val;
} bh_ini_parser_key;
typedef struct {
    int num_keys;
    char *name;
    bh_ini_parser_key *keys;
} bh_ini_parser_section;
typedef struct {
    int num_sections;
    bh_ini_parser_section *sections;
} bh_ini_parser;

static inline void bh_ini_parser_destroy(bh_ini_parser *config) {
    int i, j;
    if (!config) return;
    for (i = 0; i < config->num_sections; ++i) {
        for (j = 0; j < config->sections[i].num_keys; ++j) {
            bh_free(config->sections[i].keys[j].name);
            bh_free(config->sections[i].keys[j].val);
        }
        bh_free(config->sections[i].keys);
        bh_free(config->sections[i].name);
    }
    bh_free(config->sections);
    free(config);
}

static inline bh_ini_parser *bh_ini_parser_create(const char *filename) {
    FILE *file = fopen(filename, "r");
    bh_ini_parser *config;
    char *line;
    int ok = 1;
    if (file == NULL) {
        fprintf(stderr, "[ERROR] Could not open file: %s\n", filename);
        return NULL;
    }
    config = (bh_ini_parser *)calloc(1, sizeof(bh_ini_parser));
    while (ok && config && (line = bh_fgetline(file)) != NULL) {
        bh_strstrip(line);
        if (line[0] == '[') {
            bh_ini_parser_section *ns = (bh_ini_parser_section *)realloc(
                config->sections, (size_t)(config->num_sections + 1) * sizeof(bh_ini_parser_section));
            if (!ns) ok = 0;
            else {
                config->sections = ns;
                memset(&ns[config->num_sections], 0, sizeof(bh_ini_parser_section));
                bh_strfill(&ns[config->num_sections].name, line);
                config->num_sections++;
            }
        } else if (line[0] != '\0' && line[0] != '#' && line[0] != ';' && line[0] != '!') {
            char *eq = strchr(line, '=');
            if (config->num_sections == 0 || !eq || eq == line || eq[1] == '\0' || strchr(eq + 1, '=')) {
                fprintf(stderr, "[ERROR] Invalid key section %s\n", line);
                ok = 0;
            } else {
                bh_ini_parser_section *s = &config->sections[config->num_sections - 1];
                bh_ini_parser_key *nk =
                    (bh_ini_parser_key *)realloc(s->keys, (size_t)(s->num_keys + 1) * sizeof(bh_ini_parser_key));
                if (!nk) ok = 0;
                else {
                    s->keys = nk;
                    memset(&nk[s->num_keys], 0, sizeof(bh_ini_parser_key));
                    *eq = '\0';
                    bh_strfill(&nk[s->num_keys].name, line);
                    bh_strfill(&nk[s->num_keys].val, eq + 1);
                    s->num_keys++;
                }
            }
        }
        free(line);
    }
    fclose(file);
    if (!ok) {
        fprintf(stderr, "[ERROR] Failed to parse config file %s\n", filename);
        bh_ini_parser_destroy(config);
        return NULL;
    }
    return config;
}
#ifdef __cplusplus
}
#endif
#endif
