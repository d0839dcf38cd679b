#!/bin/bash
# stage spans of 16 batched units by unit width: rows on the dword grid (5488) against every other row off it (5490)
for w in ${WIDTHS:-5488 5490}; do echo "w=$w"; python tools/units_probe.py w=$w 6 2>&1 | grep "stage spans" | tail -1; done
