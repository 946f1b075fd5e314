#!/bin/bash
timeout 60 tools/probes/bin/valu_rate
