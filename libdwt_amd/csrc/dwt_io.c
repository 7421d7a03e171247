/*
 * dwt_io.c -- the image file readers/writers around the path:
 * ASCII PGM loaders and the text-matrix ("MAT", comma separated) reader/writer of
 * libdwt.  Host C only; images are allocated with dwt_util_alloc_image and libdwt's
 * optimal stride, exactly as the reference's loaders do, so a loaded image can go
 * straight into any transform entry.
 */
#include "../../include/libdwt.h"

#include <ctype.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char *px(void *ptr, int y, int x, int stride_x, int stride_y)
{
	return (char *)ptr + (long)y * stride_x + (long)x * stride_y;
}

/* white space and '#' comments between PGM tokens (src/libdwt.c:19384-19424) */
static int pgm_skip(FILE *f)
{
	for (;;) {
		int c = fgetc(f);
		if (c == EOF)
			return EOF;
		if (isspace(c))
			continue;
		if (c == '#') {
			do {
				c = fgetc(f);
				if (c == EOF)
					return EOF;
			} while (c != '\n');
			continue;
		}
		return ungetc(c, f) == EOF ? EOF : 0;
	}
}

/* P2 header + samples; is_float selects the element type.  Return codes as in the
 * reference (src/libdwt.c:19426-19523 float, :19525-19622 int): 1 open, 2 header,
 * 3 depth, 4 data, 5 sample out of range. */
static int load_pgm(const char *filename, int is_float, float max_s, int max_i, void **pptr, int *pstride_x, int *pstride_y,
	int *psize_x, int *psize_y)
{
	FILE *f = fopen(filename, "r");
	if (!f) {
		dwt_util_log(LOG_ERR, "Cannot open file '%s'.\n", filename);
		return 1;
	}
	int depth = 0;
	if (fgetc(f) != 'P' || fgetc(f) != '2') {
		dwt_util_log(LOG_ERR, "Invalid file header.\n");
		fclose(f);
		return 2;
	}
	int *meta[3] = {psize_x, psize_y, &depth};
	for (int k = 0; k < 3; k++) {
		pgm_skip(f);
		if (fscanf(f, "%i", meta[k]) != 1) {
			dwt_util_log(LOG_ERR, "Invalid file metadata.\n");
			fclose(f);
			return 2;
		}
	}
	if (depth >= 65536 || depth <= 0) {
		dwt_util_log(LOG_ERR, "Invalid depth.\n");
		fclose(f);
		return 3;
	}
	/* The reference computes the row pitch in `int` (4 * width, then the next prime): from 2^29 columns on that wraps, and
	 * the loop below would write the file's samples past a tiny allocation (the reference does exactly that).  The one
	 * place where this loader departs from it: such a header is refused like any other bad one. */
	if (*psize_x > (INT_MAX - 4096) / 4) {
		dwt_util_log(LOG_ERR, "Invalid file metadata.\n");
		fclose(f);
		return 2;
	}
	*pstride_y = 4;
	*pstride_x = dwt_util_get_opt_stride(*pstride_y * *psize_x);
	dwt_util_alloc_image(pptr, *pstride_x, *pstride_y, *psize_x, *psize_y);
	for (int y = 0; y < *psize_y; y++)
		for (int x = 0; x < *psize_x; x++) {
			int val;
			pgm_skip(f);
			if (fscanf(f, "%i", &val) != 1) {
				dwt_util_log(LOG_ERR, "Invalid data.\n");
				fclose(f);
				return 4;
			}
			if (val < 0 || val > depth) {
				dwt_util_log(LOG_ERR, "Invalid data depth.\n");
				fclose(f);
				return 5;
			}
			if (is_float) {
				const float v = max_s * val / depth; /* :19514 */
				memcpy(px(*pptr, y, x, *pstride_x, *pstride_y), &v, 4);
			} else {
				const int v = max_i * val / depth; /* :19613 */
				memcpy(px(*pptr, y, x, *pstride_x, *pstride_y), &v, 4);
			}
		}
	fclose(f);
	return 0;
}

int dwt_util_load_from_pgm_s(const char *filename, float max_value, void **pptr, int *pstride_x, int *pstride_y,
	int *psize_x, int *psize_y)
{
	return load_pgm(filename, 1, max_value, 0, pptr, pstride_x, pstride_y, psize_x, psize_y);
}

int dwt_util_load_from_pgm_i(const char *filename, int max_value, void **pptr, int *pstride_x, int *pstride_y,
	int *psize_x, int *psize_y)
{
	return load_pgm(filename, 0, 0.f, max_value, pptr, pstride_x, pstride_y, psize_x, psize_y);
}

/* ---- text matrices: one row per line, cells "%f" separated by ','
 * (writer src/libdwt.c:24430-24468; reader :24810-24893 with the automaton :24381-24428) ---- */
int dwt_util_save_to_mat_s(const char *path, const void *ptr, int size_x, int size_y, int stride_x, int stride_y)
{
	FILE *f = fopen(path, "w");
	if (!f)
		return 1;
	for (int y = 0; y < size_y; y++) {
		for (int x = 0; x < size_x; x++) {
			float v;
			memcpy(&v, px((void *)ptr, y, x, stride_x, stride_y), 4);
			fprintf(f, "%f", v);
			if (x + 1 != size_x)
				fputc(',', f);
		}
		fputc('\n', f);
	}
	fclose(f);
	return 0;
}

static int is_delim(int c) { return c == ',' || c == ';' || c == '\t' || c == ' '; }
static int is_newline(int c) { return c == '\n' || c == '\r'; }
/* the float reader takes 0-9 . - e + (src/libdwt.c:24838), the int reader 0-9 . - (:24918) */
static int is_number(int c, int is_float)
{
	return (c >= '0' && c <= '9') || c == '.' || c == '-' || (is_float && (c == 'e' || c == '+'));
}

/* One pass over the text.  The reader accepts cells of the characters 0-9 . - e +,
 * separated by , ; tab or space; a row is counted when a newline ends a line that held a
 * delimiter or a cell (so a last line without a newline is not a row, as in the
 * reference); the alphabet of the int reader lacks e and +; the matrix width is the smallest non-zero cell count of a line.
 * img == NULL: only count.  Returns 0, or 1 on a character outside the alphabet. */
static int mat_pass(const char *text, long n, int is_float, void *img, int stride_x, int stride_y, int size_x, int size_y,
	int *rows_out, int *min_cols_out)
{
	enum { START, DELIM, CELL } st = START;
	int rows = 0, cur = 0, min_cols = 0;
	long i = 0;
	for (;;) {
		const int c = i < n ? (unsigned char)text[i] : EOF;
		if (c == EOF || is_newline(c)) {
			/* end of line: fold this line's cell count into the minimum */
			const int a = cur ? cur : min_cols, b = min_cols ? min_cols : cur;
			min_cols = a < b ? a : b;
			cur = 0;
			if (c == EOF)
				break;
			if (st != START)
				rows++;
			st = START;
			i++;
		} else if (is_delim(c)) {
			st = DELIM;
			i++;
		} else if (is_number(c, is_float)) {
			if (st == CELL) {
				i++; /* only while counting: cells are consumed whole below */
				continue;
			}
			st = CELL;
			cur++;
			if (img) {
				char buf[256];
				int k = 0;
				while (i < n && is_number((unsigned char)text[i], is_float) && k + 1 < (int)sizeof buf)
					buf[k++] = text[i++];
				buf[k] = 0;
				float vf;
				int vi;
				const int ok = is_float ? sscanf(buf, "%f", &vf) == 1 : sscanf(buf, "%i", &vi) == 1;
				if (!ok)
					dwt_util_log(LOG_WARN, "invalid cell content\n");
				else if (cur > size_x)
					dwt_util_log(LOG_WARN, "x-coordinate is over limit\n");
				else if (rows < size_y) /* the reference writes past the image here */
					memcpy(px(img, rows, cur - 1, stride_x, stride_y), is_float ? (void *)&vf : (void *)&vi, 4);
			} else {
				i++;
			}
		} else {
			dwt_util_log(LOG_DBG, "FSM in error state: symb=%i(%c)\n", c, (char)c);
			return 1;
		}
	}
	*rows_out = rows;
	*min_cols_out = min_cols;
	return 0;
}

static int load_mat(const char *path, int is_float, void **ptr, int *size_x, int *size_y, int *stride_x, int *stride_y)
{
	*ptr = NULL;
	FILE *f = fopen(path, "r");
	if (!f)
		return 1;
	fseek(f, 0, SEEK_END);
	const long n = ftell(f);
	rewind(f);
	char *text = (char *)malloc(n > 0 ? (size_t)n : 1);
	if (!text || (n > 0 && fread(text, 1, (size_t)n, f) != (size_t)n)) {
		free(text);
		fclose(f);
		return 2;
	}
	fclose(f);
	int rows = 0, cols = 0;
	if (mat_pass(text, n, is_float, NULL, 0, 0, 0, 0, &rows, &cols)) {
		free(text);
		return 2;
	}
	if (cols > (INT_MAX - 4096) / 4) { /* (a row pitch that does not fit an int: see load_pgm) */
		free(text);
		return 2;
	}
	*size_x = cols;
	*size_y = rows;
	*stride_y = 4;
	*stride_x = dwt_util_get_opt_stride(*stride_y * *size_x);
	dwt_util_alloc_image(ptr, *stride_x, *stride_y, *size_x, *size_y);
	if (mat_pass(text, n, is_float, *ptr, *stride_x, *stride_y, cols, rows, &rows, &cols)) {
		dwt_util_free_image(ptr);
		*ptr = NULL;
		free(text);
		return 3;
	}
	free(text);
	return 0;
}

int dwt_util_load_from_mat_s(const char *path, void **ptr, int *size_x, int *size_y, int *stride_x, int *stride_y)
{
	return load_mat(path, 1, ptr, size_x, size_y, stride_x, stride_y);
}

int dwt_util_load_from_mat_i(const char *path, void **ptr, int *size_x, int *size_y, int *stride_x, int *stride_y)
{
	return load_mat(path, 0, ptr, size_x, size_y, stride_x, stride_y);
}
